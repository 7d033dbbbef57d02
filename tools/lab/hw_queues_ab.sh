#!/bin/bash
# One rank: does the number of hardware queues HIP maps the step's streams onto matter?  (bench.py sets 8 only when a process group
# exists: detector, embedder, remainder, exchange and RCCL streams; a single rank has three.)
for rep in 1 2; do for q in default 2 4 8; do
if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES', '$q', d['ms_per_step'], 'ms', d['value'], 'faces/s')"
done; done

#!/bin/bash
# Does forcing ONE workgroup per CU for one network's big kernels (lab knobs FP_PAIR_LDS_MIN / FP_X6_LDS_MIN: the launch requests
# more LDS than it uses) make the two-stream step share CUs between a BlazeFace pair kernel (vector-ALU bound) and a Depth_Wise
# kernel (matrix-core bound)?  One box, 100 timed steps, twice.
for rep in 1 2; do for cfg in "0 0" "83968 0" "0 83968" "98304 0" "0 98304"; do
set -- $cfg
FP_PAIR_LDS_MIN=$1 FP_X6_LDS_MIN=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair LDS min', '$1', 'x6 LDS min', '$2', d['ms_per_step'], 'ms', d['value'], 'faces/s')"
done; done

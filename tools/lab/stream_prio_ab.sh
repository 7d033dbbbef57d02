#!/bin/bash
# A/B of stream priorities in the two-stream step on ONE box (box-to-box variance is +-3 %): the embedder's side stream
# (FP_EMB_STREAM_PRIO) or the detector's stream (BENCH_MAIN_PRIO) at high priority (-1; the range on gfx950 is 0 .. -1).
for rep in 1 2; do for cfg in "0 " "-1 " "0 -1"; do
set -- $cfg
FP_EMB_STREAM_PRIO=$1 BENCH_MAIN_PRIO=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-fp32-leg --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('embedder prio', '$1', 'detector prio', '${2:-default}', d['ms_per_step'], 'ms', d['value'], 'faces/s')"
done; done

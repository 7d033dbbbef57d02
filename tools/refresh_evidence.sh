#!/bin/bash
# One gpurun call that regenerates every measured artefact of a round (run from the repo root on the GPU box):
#   tools/refresh_evidence.sh r02
# -> gpurun_out/<tag>_bench_* and <tag>_yolo_* (rocprofv3 stats + the two PMC passes, tools/collect_profiles.sh),
#    gpurun_out/<tag>_other_configs.jsonl (tools/config_bench.py), gpurun_out/<tag>_bench_line.json (plain bench.py,
#    CPU baseline included), gpurun_out/<tag>_planprof.log (per-op HIP-event times of the two network plans).
# Afterwards, locally: tools/profile_summary.py <tag>_bench ... / <tag>_yolo ... and copy the jsonl / json into profiles/.
set -e
tag=$1
rm -rf gpurun_out/${tag}_bench_stats gpurun_out/${tag}_bench_fetch gpurun_out/${tag}_bench_write
rm -rf gpurun_out/${tag}_yolo_stats gpurun_out/${tag}_yolo_fetch gpurun_out/${tag}_yolo_write
timeout -k 10 500 tools/collect_profiles.sh ${tag}_bench bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg
echo "[refresh] bench profiles done"
timeout -k 10 500 tools/collect_profiles.sh ${tag}_yolo tools/config_bench.py yolov5n yolov5s
echo "[refresh] yolo profiles done"
timeout -k 10 400 python3 tools/config_bench.py > gpurun_out/${tag}_other_configs.jsonl 2> gpurun_out/${tag}_other_configs.err
echo "[refresh] configs done"
timeout -k 10 300 python3 tools/plan_profile.py 256 528 > gpurun_out/${tag}_planprof.log 2>&1
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
tail -c 600 gpurun_out/${tag}_bench_line.json

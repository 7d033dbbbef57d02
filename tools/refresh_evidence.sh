#!/bin/bash
# One gpurun call that regenerates every measured artefact of a round (run from the repo root on the GPU box):
#   tools/refresh_evidence.sh r02
# -> gpurun_out/<tag>_bench_* and <tag>_yolo_* (rocprofv3 stats + the two PMC passes, tools/collect_profiles.sh),
#    gpurun_out/<tag>_other_configs.jsonl (tools/config_bench.py), gpurun_out/<tag>_bench_line.json (plain bench.py,
#    CPU baseline included), gpurun_out/<tag>_planprof_u8*.log / _yolo_planprof.log (per-op HIP-event times of the plans),
#    gpurun_out/<tag>_ablate_ops.log, _embed_split_probe.log, _split_adversarial.log, _coexec_bf16_mfma_valu.log (lab).
# Afterwards, locally: tools/profile_summary.py <tag>_bench ... / <tag>_yolo ... and copy the jsonl / json into profiles/.
set -e
tag=$1
part=${2:-all}      # "profiles" (the three rocprofv3 sets), "logs" (everything else) or "all" -- one gpurun call is 20 minutes at most
if [ "$part" != "logs" ]; then
rm -rf gpurun_out/${tag}_bench_stats gpurun_out/${tag}_bench_fetch gpurun_out/${tag}_bench_write
rm -rf gpurun_out/${tag}_yolo_stats gpurun_out/${tag}_yolo_fetch gpurun_out/${tag}_yolo_write
rm -rf gpurun_out/${tag}_bench_single_stream_stats gpurun_out/${tag}_bench_single_stream_fetch gpurun_out/${tag}_bench_single_stream_write
timeout -k 10 500 tools/collect_profiles.sh ${tag}_bench bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-other-configs
echo "[refresh] bench profiles done"
timeout -k 10 500 tools/collect_profiles.sh ${tag}_bench_single_stream bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-other-configs --overlap 0
echo "[refresh] single-stream bench profiles done"
timeout -k 10 500 tools/collect_profiles.sh ${tag}_yolo tools/config_bench.py yolov5n yolov5s
echo "[refresh] yolo profiles done"
fi
if [ "$part" = "profiles" ]; then exit 0; fi
timeout -k 10 400 python3 tools/config_bench.py > gpurun_out/${tag}_other_configs.jsonl 2> gpurun_out/${tag}_other_configs.err
echo "[refresh] configs done"
timeout -k 10 300 python3 tools/plan_profile.py 256 528 u8 > gpurun_out/${tag}_planprof_u8.log 2>&1
timeout -k 10 300 python3 tools/plan_profile.py 256 512 u8 > gpurun_out/${tag}_planprof_u8_512.log 2>&1
timeout -k 10 300 python3 tools/plan_profile.py 256 1024 u8 > gpurun_out/${tag}_planprof_u8_1024.log 2>&1
timeout -k 10 300 python3 tools/yolo_profile.py > gpurun_out/${tag}_yolo_planprof.log 2>&1
echo "[refresh] plan profiles done"
timeout -k 10 300 python3 tools/lab/ablate_ops.py "wps_kernel<48" "wps_kernel<96" "blazepair_kernel<128" "blazepair_kernel<64" stem5 "dwblock_x6_kernel<128" "dwblock_x6_kernel<64" "x6d_kernel<64, 128" blazepair_s2 stemdw > gpurun_out/${tag}_ablate_ops.log 2>&1
timeout -k 10 200 python3 tools/lab/embed_split_probe.py > gpurun_out/${tag}_embed_split_probe.log 2>&1
timeout -k 10 200 python3 tools/lab/split_adversarial_probe.py > gpurun_out/${tag}_split_adversarial.log 2>&1
timeout -k 10 100 tools/lab/coexec_bf16_lab > gpurun_out/${tag}_coexec_bf16_mfma_valu.log 2>&1
echo "[refresh] lab logs done"
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
tail -c 600 gpurun_out/${tag}_bench_line.json

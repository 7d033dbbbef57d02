#!/bin/bash
# Lab (GPU box): A/B of two complete builds of libfacepath.so on ONE box, alternating runs:
#   tools/lab/ab_libs.sh tools/lab/_ab/lib_old.so tools/lab/_ab/lib_new.so [rounds]      (AB_CMD as in ab_define.sh)
# The library in the package directory is left as the LAST one (B).
A=$1; B=$2; rounds=${3:-2}
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root" || exit 1
for r in $(seq 1 "$rounds"); do
  for v in "$A" "$B"; do
    cp "$v" face_detection_and_recognition_amd/libfacepath.so
    echo "[$v]"
    if [ -n "$AB_CMD" ]; then bash -c "$AB_CMD"; continue; fi
    python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-other-configs --no-fp32-leg 2>/dev/null |
       python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms two-stream;', d['roofline']['kernel'], d['roofline']['avg_launch_us'], 'us beside /', d['roofline']['exclusive_avg_launch_us'], 'us alone')"
  done
done

#!/bin/bash
# Lab (GPU box): A/B of one csrc/*.hip file built with two sets of defines, bench.py's timed step, alternating runs on ONE box.
#   tools/lab/ab_define.sh blazepair.hip "-DFP_PAIR_DIRECT_STORE=0" "-DFP_PAIR_DIRECT_STORE=1" [rounds]
# AB_CMD=<shell command run from the repo root> replaces the bench run (e.g. a plan_profile line filter).
src=$1; A=$2; B=$3; rounds=${4:-3}
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root/face_detection_and_recognition_amd/csrc" || exit 1
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
obj="${src%.hip}.o"
for defs in "$A" "$B"; do
  /opt/rocm/bin/hipcc $FL $defs -c "$src" -o "/tmp/ab_$(echo "$defs" | md5sum | cut -c1-8).o" || exit 1
done
for r in $(seq 1 "$rounds"); do
  for defs in "$A" "$B"; do
    cp "/tmp/ab_$(echo "$defs" | md5sum | cut -c1-8).o" "$obj"
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libfacepath.so *.o || exit 1
    if [ -n "$AB_CMD" ]; then echo "[$defs]"; (cd "$root" && bash -c "$AB_CMD"); continue; fi
    (cd "$root" && python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-other-configs --no-fp32-leg 2>/dev/null |
       python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$defs]', d['ms_per_step'], 'ms two-stream;', d['roofline']['kernel'], d['roofline']['avg_launch_us'], 'us beside /', d['roofline']['exclusive_avg_launch_us'], 'us alone')")
  done
done

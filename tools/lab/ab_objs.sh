#!/bin/bash
# Lab (GPU box): A/B of two prebuilt objects of one csrc/*.hip file (e.g. the file at HEAD against the working tree), alternating
# runs on ONE box:   tools/lab/ab_objs.sh dwblockx6.o tools/lab/_ab/dwblockx6_old.o tools/lab/_ab/dwblockx6_new.o [rounds]
# AB_CMD=<shell command run from the repo root> replaces the bench run.  The library is left linked with the LAST object (B).
obj=$1; A=$2; B=$3; rounds=${4:-2}
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root/face_detection_and_recognition_amd/csrc" || exit 1
for r in $(seq 1 "$rounds"); do
  for v in "$A" "$B"; do
    cp "$root/$v" "$obj"
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libfacepath.so *.o || exit 1
    echo "[$v]"
    if [ -n "$AB_CMD" ]; then (cd "$root" && bash -c "$AB_CMD"); continue; fi
    (cd "$root" && python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-other-configs --no-fp32-leg 2>/dev/null |
       python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms two-stream;', d['roofline']['kernel'], d['roofline']['avg_launch_us'], 'us beside /', d['roofline']['exclusive_avg_launch_us'], 'us alone')")
  done
done

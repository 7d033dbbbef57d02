#!/bin/bash
# Lab (GPU box): several prebuilt objects of one csrc/*.hip file in turn on ONE box (ablation builds: wrong results, timing only):
#   AB_CMD=... tools/lab/ab_many.sh stem.o tools/lab/_ab/stem_abl0.o tools/lab/_ab/stem_abl1.o ...
# The library is left linked with the FIRST object.
obj=$1; shift
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root/face_detection_and_recognition_amd/csrc" || exit 1
for v in "$@" "$1"; do
  cp "$root/$v" "$obj"
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libfacepath.so *.o || exit 1
  echo "[$v]"
  (cd "$root" && bash -c "$AB_CMD")
done

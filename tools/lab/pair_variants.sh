#!/bin/bash
# Lab (GPU box): rebuild csrc/blazepair.hip with each set of defines, relink the library IN THE BOX'S SCRATCH COPY and time the
# detector plan's blazepair launches (tools/plan_profile.py).  Usage: tools/lab/pair_variants.sh "" "-DFP_PAIR_ABLATE=16" ...
cd "$(dirname "$0")/../../face_detection_and_recognition_amd/csrc" || exit 1
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
for defs in "$@" ""; do
  /opt/rocm/bin/hipcc $FL $defs -c blazepair.hip -o blazepair.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libfacepath.so *.o || exit 1
  echo "== blazepair.hip built with [$defs]"
  (cd ../.. && python tools/plan_profile.py 256 512 u8 2>/dev/null | grep -E "blazepair_kernel|blazeface-back" )
done

#!/bin/bash
# Rebuild libfacepath.so and the x6 stamp lab (run from the repo root).
set -e
make -C face_detection_and_recognition_amd/csrc 2>&1 | grep -E "error|warning" -A3 || true
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFP_X6_STAMPS -Iinclude -Iface_detection_and_recognition_amd/csrc \
      tools/lab/x6_lab.hip -o tools/lab/x6_lab -Wno-unused-value -Wno-unused-result 2>&1 | grep -E "error" -A5 || true

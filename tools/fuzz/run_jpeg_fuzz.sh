#!/bin/bash
# Mutation-fuzzes the HOST half of the JPEG decoder (fp_jpeg_parse + fp_jpeg_entropy_decode, csrc/jpeg.hip) under
# AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only: jpeg.hip is compiled --cuda-host-only, the device code object is
# replaced by an empty blob (no kernel is ever launched), nothing touches a GPU.
#     tools/fuzz/run_jpeg_fuzz.sh [iterations per seed file = 2000] [PRNG seed = 1] [seed files ... = generated with Pillow]
# Prints "<n> inputs: <a> decoded, <b> refused" and exits 0 when no sanitizer report came; a report aborts with exit != 0.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
ITERS="${1:-2000}"; SEED="${2:-1}"; shift $(( $# > 2 ? 2 : $# ))
OUT="${FUZZ_BUILD_DIR:-$(mktemp -d /tmp/jpeg_fuzz.XXXXXX)}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CXX="${FUZZ_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
"$HIPCC" -O1 -g -std=c++17 --cuda-host-only -x hip $SAN -I"$ROOT/include" -c "$ROOT/face_detection_and_recognition_amd/csrc/jpeg.hip" -o "$OUT/jpeg_host.o" 2>/dev/null
SYM="$(nm "$OUT/jpeg_host.o" | awk '/__hip_fatbin_/ {print $NF; exit}')"
cat > "$OUT/stubs.cpp" <<EOS
#include <hip/hip_runtime_api.h>
extern "C" { __attribute__((aligned(4096))) extern const char $SYM[4096]; const char $SYM[4096] = {0}; }   // the device code object
void fp_set_hip_error(hipError_t) {}
EOS
"$CXX" -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$OUT/stubs.cpp" -o "$OUT/stubs.o"
"$CXX" -O1 -g $SAN -I"$ROOT/include" -c "$ROOT/tools/fuzz/jpeg_fuzz.cpp" -o "$OUT/fuzz.o"
"$CXX" $SAN "$OUT/fuzz.o" "$OUT/jpeg_host.o" "$OUT/stubs.o" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o "$OUT/fuzz"
if [ $# -eq 0 ]; then
  python3 - "$OUT" <<'EOS'
import io, os, sys
import numpy as np
from PIL import Image
out = os.path.join(sys.argv[1], "seeds")
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(0)
i = 0
for (w, h) in ((64, 48), (67, 45), (17, 9), (1, 1)):
    for sub in (0, 1, 2):
        for kw in (dict(quality=30), dict(quality=92, restart_marker_blocks=3), dict(quality=40, progressive=True),
                   dict(quality=95, progressive=True, restart_marker_blocks=2)):
            img = np.clip(np.cumsum(np.cumsum(rng.normal(0, 3, (h, w, 3)), 0), 1) + 128, 0, 255).astype(np.uint8)
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", subsampling=sub, **kw)
            open(os.path.join(out, f"s{i:02d}.jpg"), "wb").write(b.getvalue())
            i += 1
g = np.clip(rng.normal(128, 40, (33, 70)), 0, 255).astype(np.uint8)
for kw in (dict(quality=75), dict(quality=75, progressive=True)):
    b = io.BytesIO()
    Image.fromarray(g).save(b, "JPEG", **kw)
    open(os.path.join(out, f"s{i:02d}.jpg"), "wb").write(b.getvalue())
    i += 1
EOS
  set -- "$OUT"/seeds/*.jpg
fi
FUZZ_SEED="$SEED" "$OUT/fuzz" "$ITERS" "$@"

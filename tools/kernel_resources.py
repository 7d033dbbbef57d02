#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS / occupancy table of one csrc/*.hip file
(hipcc -Rpass-analysis=kernel-resource-usage):   python tools/kernel_resources.py dwblockx6.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "face_detection_and_recognition_amd", "csrc")
PATS = (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"))


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    with tempfile.NamedTemporaryFile(suffix=".o") as tmp:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                              "-Rpass-analysis=kernel-resource-usage"] + extra + ["-c", src, "-o", tmp.name],
                             cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in PATS:
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    if not rows:                       # the compile failed: show why (c++filt without arguments would wait on stdin)
        sys.stderr.write(out[-3000:])
        return 1
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True,
                           text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
        print(f"{n:64s} vgpr {r.get('vgpr', 0):3d} agpr {r.get('agpr', 0):3d} spill {r.get('spill', 0):3d} "
              f"scratch {r.get('scratch', 0):4d} lds {r.get('lds', 0):6d} occ {r.get('occ', 0)}")


if __name__ == "__main__":
    main()

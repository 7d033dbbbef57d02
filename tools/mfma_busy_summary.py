"""Matrix-pipe utilisation per kernel family from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass (tools/mfma_counters.sh):
busy cycles summed over the SIMDs / (kernel duration x 2.4 GHz peak clock x 1024 SIMDs).  The counter advances 16 per
v_mfma_f32_16x16x32_bf16 and 64 per v_mfma_f32_32x32x2_f32 (pass counts x 4).   usage: mfma_busy_summary.py <pmc dir>"""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def main():
    d = sys.argv[1]
    busy = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                b = busy[short(r["Kernel_Name"])]
                b[0] += float(r["Counter_Value"])
                b[1] += 1
    dur = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            t = dur[short(r["Kernel_Name"])]
            t[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            t[1] += 1
    print("| kernel | launches | avg us | MFMA busy cycles / launch | matrix-pipe utilisation (of 1024 SIMDs x 2.4 GHz) |")
    print("|---|---|---|---|---|")
    rows = []
    for k, (b, n) in busy.items():
        if b <= 0 or k not in dur:
            continue
        us = dur[k][0] / dur[k][1] / 1e3
        per = b / n
        rows.append((per / (us * 1e-6 * 2.4e9 * 1024), k, n, us, per))
    for u, k, n, us, per in sorted(rows, reverse=True):
        print(f"| `{k}` | {n} | {us:.1f} | {per:.3g} | {100 * u:.1f} % |")


if __name__ == "__main__":
    main()

"""Per-kernel averages of a rocprofv3 --pmc pass (counter_collection CSV): tools/sq_counters.py <dir> [<dir> ...]
Prints one row per kernel family with every counter found, per launch, plus the ratios used in profiles/*_sq_counters.md
(wait / issue-stall / issuing fractions of SQ_WAVE_CYCLES, VALU and MFMA instructions per wave)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for k in sorted(agg):
        c = {n: v[0] / max(v[1], 1) for n, v in agg[k].items()}
        n = max(v[1] for v in agg[k].values())
        line = [f"{k[:44]:44s} n={n:4d}"]
        wc = c.get("SQ_WAVE_CYCLES")
        w = c.get("SQ_WAVES")
        if wc and w:
            line.append(f"waves {w:8.0f} cyc/wave*4 {4 * wc / w:9.0f}")
            for nm, lab in (("SQ_WAIT_ANY", "wait"), ("SQ_WAIT_INST_ANY", "issue-stall"), ("SQ_ACTIVE_INST_ANY", "issuing")):
                if nm in c:
                    line.append(f"{lab} {c[nm] / wc:.2f}")
            for nm, lab in (("SQ_INSTS_VALU", "VALU/wave"), ("SQ_INSTS_MFMA", "MFMA/wave"), ("SQ_INSTS_LDS", "LDS/wave"),
                            ("SQ_INSTS_VMEM_RD", "VMEMrd/wave"), ("SQ_INSTS_SALU", "SALU/wave")):
                if nm in c:
                    line.append(f"{lab} {c[nm] / w:.0f}")
        for nm in sorted(c):
            if nm not in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU",
                          "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU"):
                line.append(f"{nm} {c[nm]:.3g}")
        print("  ".join(line))


if __name__ == "__main__":
    main()

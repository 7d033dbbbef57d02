"""Per-kernel averages of the derived PMC metrics collected by tools/pmc_passes.sh (one directory per rocprofv3 pass).
usage: pmc_summary.py <pmc dir> [<pmc dir> ...]   -> markdown table on stdout"""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def main():
    vals = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    names = []
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k, c = short(r["Kernel_Name"]), r["Counter_Name"]
                if c not in names:
                    names.append(c)
                v = vals[k][c]
                v[0] += float(r["Counter_Value"])
                v[1] += 1
    keep = [k for k in vals if not k.startswith(("at::", "__amd", "void at::")) and "elementwise" not in k and "Fill" not in k]
    print("| kernel | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    for k in sorted(keep):
        row = []
        for c in names:
            s, n = vals[k][c]
            row.append(f"{s / n:.4g}" if n else "")
        print(f"| `{k}` | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()

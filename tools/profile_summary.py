"""Condense rocprofv3 outputs (kernel stats CSV + the two PMC passes) into profiles/<tag>_*.

  python tools/profile_summary.py r01 gpurun_out/prof_bench gpurun_out/pmc_fetch gpurun_out/pmc_write

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --kernel-trace --stats summary),
profiles/<tag>_pmc_traffic.json (per kernel: launches, FETCH_SIZE/WRITE_SIZE averages and the HBM bytes per launch
with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on wide coalesced
reads -> x2; counters are in KiB) and profiles/<tag>_summary.md.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    cmd = sys.argv[5] if len(sys.argv) > 5 else "bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg"
    os.makedirs("profiles", exist_ok=True)
    ks = glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv"))[0]
    shutil.copy(ks, f"profiles/{tag}_kernel_stats.csv")
    stats = list(csv.DictReader(open(ks)))

    def pmc(d):
        """kernel -> [counter sum, launches] of the kernel's LARGEST launches (by grid size): since round 4 the embedder's
        kernels are launched twice per step -- on the first 512 crops and on the remainder of ~16 (pipeline._embed_split) -- and
        a per-launch figure must not average the two."""
        f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
        groups = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            a = groups[(short(r["Kernel_Name"]), int(r["Grid_Size"]))]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        agg = {}
        for (name, grid), a in groups.items():
            if name not in agg or grid > agg[name][2]:
                agg[name] = [a[0], a[1], grid]
        return agg

    fe, wr = pmc(fetch_dir), pmc(write_dir)
    traffic = {}
    for k in fe:
        if k in wr and fe[k][1] and wr[k][1]:
            f_kib, w_kib = fe[k][0] / fe[k][1], wr[k][0] / wr[k][1]
            traffic[k] = {"launches_profiled": fe[k][1], "grid_size": fe[k][2], "FETCH_SIZE_KiB_avg": round(f_kib, 1),
                          "WRITE_SIZE_KiB_avg": round(w_kib, 1),
                          "hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024)}
    json.dump(traffic, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
    with open(f"profiles/{tag}_summary.md", "w") as o:
        o.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 {cmd}`\n\n")
        o.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in stats[:24]:
            o.write(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
        o.write("\n`at::native::*copy*` / `__amd_rocclr_copyBuffer` / `fillBuffer*` rows with thousands of calls are SETUP (the synthetic "
                "weights and frame batches uploaded tensor by tensor before the first step), not part of a step: a step launches "
                "about 70 kernels, four of them torch's.\n")
        o.write("\nThe embedder's kernels appear twice per step: on the first 512 crops and, on a side stream, on the remainder of "
                "~16 crops (pipeline._embed_split); `avg us` above averages both kinds of launch.\n")
        o.write("\nPMC (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, KiB, per launch average over a kernel's "
                "LARGEST-grid launches; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, the x2 being the gfx950 FETCH_SIZE correction):\n\n")
        o.write("| kernel | FETCH KiB | WRITE KiB | HBM MB / launch |\n|---|---|---|---|\n")
        for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"])[:16]:
            o.write(f"| `{k}` | {v['FETCH_SIZE_KiB_avg']} | {v['WRITE_SIZE_KiB_avg']} | {v['hbm_bytes_per_launch'] / 1e6:.1f} |\n")
    print(open(f"profiles/{tag}_summary.md").read())


if __name__ == "__main__":
    main()

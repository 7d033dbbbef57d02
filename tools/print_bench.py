"""One-line view of a bench.py JSON line: python tools/print_bench.py <file> [label]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(sys.argv[2] if len(sys.argv) > 2 else "", d["value"], d["unit"], d["ms_per_step"], "ms/step", d["config"].get("frames_per_s"),
      "frames/s |", r.get("kernel"), r.get("avg_launch_us"), "us x", r.get("launches_per_step"), "frac", r.get("frac"))

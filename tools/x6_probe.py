"""Single Depth_Wise block timing: the bf16x6 split-MFMA kernel (csrc/dwblockx6.hip) against the fp32-MFMA forms, per batch.
usage: python tools/x6_probe.py [hw:N ...]   (hw in 14 / 28 / 7; default: the bench-like sizes)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import Depth_Wise  # noqa: E402
from face_detection_and_recognition_amd.plan import CompiledPlan, PlanBuilder  # noqa: E402
from face_detection_and_recognition_amd.synth import synth_state_dict  # noqa: E402


def build(hw, n, x6, dev, stride=1):
    cin = {28: 64, 14: 128, 7: 128, 56: 64}[hw]
    cout, groups = (cin, 2 * cin) if stride == 1 or hw == 56 else (128, 4 * cin)
    blk = Depth_Wise(cin, cout, residual=stride == 1, kernel=(3, 3), stride=(stride, stride), padding=(1, 1), groups=groups)
    blk.load_state_dict(synth_state_dict(blk.state_dict(), 5))
    pb = PlanBuilder(n)
    inp = pb.new_buf(hw, hw, cin)
    old = Depth_Wise.X6
    Depth_Wise.X6 = x6
    try:
        blk.emit(pb, inp.view())
    finally:
        Depth_Wise.X6 = old
    plan = CompiledPlan(pb, dev)
    plan.buf_tensor(inp, n).normal_()
    return plan


def time_plan(plan, reps=30):
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    args = sys.argv[1:] or ["14:128", "14:256", "14:264", "14:384", "14:512", "14:528", "14:1024", "28:64", "28:128", "28:528", "28:1024", "7:528", "7:1024"]
    for a in args:
        stride = 2 if a.startswith("s2:") else 1
        hw, n = (int(v) for v in a.replace("s2:", "").split(":"))
        cin = {28: 64, 14: 128, 7: 128, 56: 64}[hw]
        flop = 2.0 * n * hw * hw * cin * 2 * cin * 2 if stride == 1 else 2.0 * n * (hw * hw * cin * 4 * cin + hw * hw / 4 * 4 * cin * 128)
        row = []
        for x6 in (True, False):
            p = build(hw, n, x6, dev, stride)
            us = time_plan(p)
            names = "+".join(p.kernel_name(i).split("<")[0] for i in range(p.n_ops))
            row.append(f"{names} {us:7.1f} us {flop / us / 1e6:6.1f} TF/s")
        print(f"hw={hw} N={n:5d}: " + "   |   ".join(row), flush=True)


if __name__ == "__main__":
    main()

"""Lab: does a 1x1 conv that reads one 64-channel half of a 128-channel NHWC tensor (ShuffleV2 branch2's first conv,
y5/models/common.py:169-172) run slower than the same conv on a dense 64-channel tensor?  (r02: yes, 300 vs 192 us.)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_detection_and_recognition_amd import _lib as L  # noqa: E402
from face_detection_and_recognition_amd.plan import CompiledPlan, PlanBuilder, View  # noqa: E402


def run(N, HW, cin_buf, coff, cin, cout, out_buf_c, out_coff, reps=20):
    pb = PlanBuilder(N)
    x = pb.new_buf(HW, HW, cin_buf)
    y = pb.new_buf(HW, HW, out_buf_c)
    w = np.random.default_rng(0).normal(0, 0.1, (cout, cin, 1, 1)).astype(np.float32)
    pb.conv(View(x, coff, cin), w, View(y, out_coff, cout), bias=np.zeros(cout, np.float32), act=L.ACT_SILU)
    plan = CompiledPlan(pb, "cuda:0")
    plan.arena.normal_()
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        plan.run()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3, plan.kernel_name(0)


if __name__ == "__main__":
    for HW, C in ((80, 64), (40, 128), (20, 256)):
        for tag, args in (("dense in, dense out", (C, 0, C, C, C, 0)),
                          ("upper half of 2C in", (2 * C, C, C, C, C, 0)),
                          ("lower half of 2C in", (2 * C, 0, C, C, C, 0)),
                          ("dense in, half of 2C out", (C, 0, C, C, 2 * C, C)),
                          ("full 2C in", (2 * C, 0, 2 * C, C, C, 0))):
            us, name = run(256, HW, *args)
            print(f"{HW}x{HW} C={C} {tag:28s} {us:8.1f} us  {name}", flush=True)

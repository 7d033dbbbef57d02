"""Per-op timing of the two network plans with the timed executor (HIP events on the launch stream)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd import _lib as L  # noqa: E402


def profile(plan, name, reps=10):
    mask = bytes([1] * plan.n_ops)
    ms = (ctypes.c_float * plan.n_ops)()
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    timers = [plan.new_timer() for _ in range(reps)]
    for t in timers:
        plan.run_timed(t, mask)
    torch.cuda.synchronize()
    for t in timers:
        plan.accumulate(t, ms)
        plan.destroy_timer(t)
    tot = sum(ms) / reps
    print(f"== {name}: {plan.n_ops} ops, {tot:.3f} ms/run")
    for i in range(plan.n_ops):
        op = plan.ops[i]
        t = ms[i] / reps
        gb = plan.algorithmic_bytes(i) / 1e9
        fl = plan.flops(i)
        print(f"{i:3d} {plan.kernel_name(i):34s} {op.H:4d}x{op.W:<4d} {op.Cin:4d}->{op.Cout:<4d} k{op.KH} s{op.stride} "
              f"{t * 1e3:9.1f} us {100 * t / tot:5.1f}%  {gb / (t * 1e-3 + 1e-12):8.0f} GB/s {fl / (t * 1e-3 + 1e-12) / 1e12:6.1f} TF/s")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    NF = int(sys.argv[2]) if len(sys.argv) > 2 else 1088
    dev = torch.device("cuda:0")
    det = W.build_blazeface_back(dev)
    emb = W.build_embedder(dev)
    if len(sys.argv) > 3 and sys.argv[3] == "u8":      # the bench's form: the stem reads the u8 frames (letterbox inside)
        from face_detection_and_recognition_amd.modules.utils.image import bind_letterbox
        frames = W.make_frames(B, dev, seed=1234)
        p = det.plan_for(B, frame_hw=tuple(frames.shape[1:3]))
        bind_letterbox(p, frames, det._preprocess_lut(), pad_value=125, swap_rb=True)
        profile(p, f"blazeface-back B={B}, u8 frames {tuple(frames.shape[1:3])}")
    else:
        p = det.plan_for(B)
        p.input.normal_()
        profile(p, f"blazeface-back B={B}")
    q = emb.plan_for(NF)
    q.input.normal_()
    profile(q, f"mobilefacenet N={NF}")


if __name__ == "__main__":
    main()

"""Does running two copies of a network plan on two streams beat running them back to back?  (Would a third stream pay?)
usage: python tools/lab/concurrency_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd.modules.utils.image import bind_letterbox  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    frames = [W.make_frames(256, dev, seed=1234 + i) for i in range(2)]
    dets = [W.build_blazeface_back(dev) for _ in range(2)]
    for d in dets:
        d.co_scheduled = True
    plans = []
    for d, f in zip(dets, frames):
        p = d.plan_for(256, frame_hw=tuple(f.shape[1:3]))
        bind_letterbox(p, f, d._preprocess_lut(), pad_value=125, swap_rb=True)
        plans.append(p)
    embs = [W.build_embedder(dev) for _ in range(2)]
    eplans = [e.plan_for(528) for e in embs]
    for q in eplans:
        q.input.normal_()
    s = [torch.cuda.Stream(device=dev) for _ in range(4)]

    def seq(ps):
        for p in ps:
            p.run()

    def par(ps):
        ev = torch.cuda.Event()
        ev.record()
        for p, st in zip(ps, s):
            with torch.cuda.stream(st):
                st.wait_event(ev)
                p.run()
        for st in s[:len(ps)]:
            torch.cuda.current_stream().wait_stream(st)

    for name, ps in (("detector x2", plans), ("embedder x2", eplans), ("detector + embedder", [plans[0], eplans[0]]),
                     ("2 detectors + 2 embedders", plans + eplans)):
        a, b = timed(lambda: seq(ps)), timed(lambda: par(ps))
        print(f"{name}: back to back {a:.3f} ms, two streams {b:.3f} ms ({a / b:.3f}x)", flush=True)


if __name__ == "__main__":
    main()

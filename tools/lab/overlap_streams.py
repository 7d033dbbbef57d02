"""Lab: does running detect(k+1) on a second stream while embed(k) runs buy anything?  (tails of ~120 kernels per
step + the one host sync per step).  Prints sequential vs overlapped ms/step for the bench workload."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd.pipeline import FacePipeline  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, K = 256, 20
    frames = W.make_frames(B, dev)
    det = W.build_detector(dev, frames[:64])
    emb = W.build_embedder(dev)
    pipe = FacePipeline(det, emb, W.make_reference(10000, dev), tau=0.3)
    for _ in range(3):
        pipe.step(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        pipe.step(frames)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / K * 1e3

    sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    host_n = torch.zeros((1,), dtype=torch.int32).pin_memory()

    def front(k):
        with torch.cuda.stream(sA):
            dets, counts = pipe.detect(frames)
            items, info, nf = pipe.crops(frames, dets, counts)
            hn = torch.empty((1,), dtype=torch.int32).pin_memory()
            hn.copy_(nf, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(sA)
        return items, info, hn, ev

    def back(items, info, hn, ev):
        ev.synchronize()
        n = int(hn[0])
        with torch.cuda.stream(sB):
            sB.wait_event(ev)
            items.record_stream(sB)
            e = pipe.embed(frames, items, n)
            pipe.filter(e)
        return n

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cur = front(0)
    tot = 0
    for k in range(K):
        nxt = front(k + 1) if k + 1 < K else None
        tot += back(*cur)
        cur = nxt
    torch.cuda.synchronize()
    ov = (time.perf_counter() - t0) / K * 1e3
    print(f"sequential {seq:.3f} ms/step, two streams {ov:.3f} ms/step ({tot // K} faces/step)")


if __name__ == "__main__":
    main()

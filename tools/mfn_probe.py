"""Mobile-FaceNet per-op timing with and without the whole-block kernels (FP_OP_DWBLOCK), same process, interleaved.
usage: python tools/mfn_probe.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import Depth_Wise  # noqa: E402
from tools.plan_profile import profile  # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:]] or [528, 1024]
    dev = torch.device("cuda:0")
    emb = W.build_embedder(dev)
    for n in ns:
        plans = {}
        x = torch.randn((n, 112, 112, 4), device=dev)
        variants = {True: (14, 7), False: (), "14": (14,), "7": (7,)}
        for fb, shapes in variants.items():
            plans[fb] = emb._build(n, block_shapes=shapes)
            plans[fb].input.copy_(x)
        for rep in range(2):
            for fb in (True, False):
                if rep == 1:
                    profile(plans[fb], f"mobilefacenet N={n} whole-block={fb}")
                else:
                    for _ in range(3):
                        plans[fb].run()
        # end-to-end wall per forward, interleaved
        for fb in (True, False, "14", "7", True, False):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plans[fb].run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"## N={n} whole-block={fb}: {ms:.3f} ms/forward = {n / ms:.1f} k crops/s", flush=True)
        d = (plans[True].out - plans[False].out).abs().max().item()
        print(f"## N={n} max |emb(whole-block) - emb(two-launch)| = {d:.3e}", flush=True)


if __name__ == "__main__":
    main()

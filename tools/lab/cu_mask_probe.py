"""Lab: does giving the detector and the embedder their own halves of the chip (hipExtStreamCreateWithCUMask) beat letting the
dispatcher mix their workgroups on every CU?  One detector plan (256 frames) and one embedder plan (512 crops), enqueued on two
streams and timed together: plain streams; masks = lower / upper half of the 256 CU bits; masks = even / odd bits.
usage: python tools/lab/cu_mask_probe.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd.modules.utils.image import bind_letterbox  # noqa: E402


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits(32 * w + b)) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.cuda.init()
    hip = ctypes.CDLL("libamdhip64.so")
    frames = W.make_frames(256, dev, seed=1234)
    det = W.build_blazeface_back(dev)
    det.co_scheduled = True
    dp = det.plan_for(256, frame_hw=tuple(frames.shape[1:3]))
    bind_letterbox(dp, frames, det._preprocess_lut(), pad_value=125, swap_rb=True)
    ep = W.build_embedder(dev).plan_for(512)
    ep.input.normal_()

    def run_on(sa, sb):
        ev = torch.cuda.Event()
        ev.record()
        for p, st in ((dp, sa), (ep, sb)):
            with torch.cuda.stream(st):
                st.wait_event(ev)
                p.run()
        torch.cuda.current_stream().wait_stream(sa)
        torch.cuda.current_stream().wait_stream(sb)

    a = timed(lambda: (dp.run(), ep.run()))
    print(f"back to back on one stream: {a:.3f} ms", flush=True)
    cases = {"plain streams": (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)),
             "lower / upper half of the mask bits": (masked_stream(hip, lambda i: i < 128), masked_stream(hip, lambda i: i >= 128)),
             "even / odd mask bits": (masked_stream(hip, lambda i: i % 2 == 0), masked_stream(hip, lambda i: i % 2 == 1)),
             "bits 0-3 / 4-7 of every eight": (masked_stream(hip, lambda i: i % 8 < 4), masked_stream(hip, lambda i: i % 8 >= 4))}
    for rep in range(2):
        for name, (sa, sb) in cases.items():
            print(f"{name}: {timed(lambda: run_on(sa, sb)):.3f} ms", flush=True)
    for name, (sa, sb) in list(cases.items())[1:]:
        with torch.cuda.stream(sa):
            t = timed(lambda: dp.run())
        with torch.cuda.stream(sb):
            u = timed(lambda: ep.run())
        print(f"{name}: detector alone on its part {t:.3f} ms, embedder alone on its part {u:.3f} ms", flush=True)


if __name__ == "__main__":
    main()

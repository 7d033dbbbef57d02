"""Where a batch JPEG decode's time goes: host Huffman on 1 / 16 threads (with and without a pinned target), the copy, the
device half.    python tools/lab/jpeg_stage_probe.py"""
import ctypes as C
import io
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from face_detection_and_recognition_amd import _lib as L  # noqa: E402
from face_detection_and_recognition_amd.modules.utils import jpeg as J  # noqa: E402
from PIL import Image  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
img = np.clip(np.cumsum(np.cumsum(rng.normal(0, 2.5, (576, 1024, 3)), 0), 1) * 0.2 + rng.normal(128, 20, (576, 1024, 3)), 0, 255).astype(np.uint8)
b = io.BytesIO()
Image.fromarray(img).save(b, "JPEG", quality=90, subsampling=2)
data = b.getvalue()
n = 256
lib = L.load()
info, buf = J.parse(data)
tgt = torch.empty((int(info.n_coefs),), dtype=torch.int16)
pin = torch.empty((int(info.n_coefs),), dtype=torch.int16, pin_memory=True)


def raw(t):
    i2 = L.FpJpegInfo()
    lib.fp_jpeg_parse(buf, len(data), C.byref(i2))
    lib.fp_jpeg_entropy_decode(buf, len(data), C.byref(i2), C.c_void_p(t.data_ptr()))


def clock(name, fn, reps=1):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt * 1e3:.2f} ms", flush=True)


clock("C Huffman, one call, pageable target", lambda: raw(tgt), 20)
clock("C Huffman, one call, pinned target", lambda: raw(pin), 20)
clock("entropy_decode (python wrapper, pinned alloc)", lambda: J.entropy_decode(data, pinned=True), 20)
clock("entropy_decode (python wrapper, pageable alloc)", lambda: J.entropy_decode(data, pinned=False), 20)
for th in (1, 4, 8, 16):
    tg = [torch.empty_like(tgt) for _ in range(th)]
    with ThreadPoolExecutor(th) as pool:
        clock(f"{n} raw C Huffman calls on {th} threads (reused targets)",
              lambda: list(pool.map(lambda i: raw(tg[i % th]), range(n))))
with ThreadPoolExecutor(16) as pool:
    clock(f"{n} entropy_decode(pinned) on 16 threads", lambda: list(pool.map(lambda i: J.entropy_decode(data, pinned=True), range(n))))
    clock(f"{n} entropy_decode(pageable) on 16 threads", lambda: list(pool.map(lambda i: J.entropy_decode(data, pinned=False), range(n))))
cd = pin.to(dev)
clock("H2D of one frame's coefficients (pinned)", lambda: cd.copy_(pin, non_blocking=True), 50)
out = J.reconstruct(info, cd, dev)
clock("device half", lambda: J.reconstruct(info, cd, dev, out=out), 50)
clock(f"decode_jpeg_batch {n} frames, 16 threads", lambda: J.decode_jpeg_batch([data] * n, dev, threads=16))

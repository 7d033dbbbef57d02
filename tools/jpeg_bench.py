"""Decode rate of the JPEG front end (modules/utils/jpeg.py) against Pillow on the host: 256 synthetic 576 x 1024 4:2:0 frames.
    python tools/jpeg_bench.py [n_frames] [threads]"""
import io
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_detection_and_recognition_amd.modules.utils import jpeg as J  # noqa: E402
from PIL import Image  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
datas = []
for i in range(8):
    img = np.clip(np.cumsum(np.cumsum(rng.normal(0, 2.5, (576, 1024, 3)), 0), 1) * 0.2 + rng.normal(128, 20, (576, 1024, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=90, subsampling=2)
    datas.append(b.getvalue())
datas = [datas[i % 8] for i in range(n)]
print(f"{n} frames 576x1024 4:2:0 q90, {sum(len(d) for d in datas) / n / 1024:.0f} KiB each, {threads} host threads")


def pil_one(d):
    return np.asarray(Image.open(io.BytesIO(d)).convert("RGB"))


for name, fn in (("Pillow decode on the host + upload", lambda: torch.from_numpy(np.stack(list(ThreadPoolExecutor(threads).map(pil_one, datas)))).to(dev)),
                 ("host Huffman + device reconstruction", lambda: torch.stack(J.decode_jpeg_batch(datas, dev, threads=threads)))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3:.1f} ms = {n / dt:.0f} frames/s  ({tuple(out.shape)})")
# the device half alone
info, coefs = J.entropy_decode(datas[0], pinned=True)
cd = coefs.to(dev)
out = J.reconstruct(info, cd, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    J.reconstruct(info, cd, dev, out=out)
torch.cuda.synchronize()
print(f"device half alone: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per frame")
t0 = time.perf_counter()
for _ in range(20):
    J.entropy_decode(datas[0])
print(f"host half alone (one thread): {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per frame")

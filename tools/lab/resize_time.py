import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from face_detection_and_recognition_amd import _lib as L
dev = torch.device("cuda:0")
lib = L.load()
rng = np.random.default_rng(0)
frames = torch.from_numpy(rng.integers(0, 256, (256, 576, 1024, 3), dtype=np.uint8)).to(dev)
n = 528
items = np.zeros((n, 9), np.int32)
for k in range(n):
    w = int(rng.integers(60, 300)); h = int(rng.integers(60, 300))
    items[k] = [k % 256, int(rng.integers(0, 1024 - w)), int(rng.integers(0, 576 - h)), w, h, 0, 0, 112, 112]
it = torch.from_numpy(items).to(dev)
lut = torch.linspace(-1, 1, 256, device=dev)
canvas = torch.empty((n, 112, 112, 4), device=dev)
for mode in ("", "1", "", "1"):
    if mode: os.environ["FP_RESIZE_PER_PIXEL"] = "1"
    else: os.environ.pop("FP_RESIZE_PER_PIXEL", None)
    lib.fp_debug_reload_env()
    for _ in range(3):
        lib.fp_resize_normalize(L.ptr(frames), 256, 576, 1024, L.ptr(it), n, L.ptr(canvas), 112, 112, 4, L.ptr(lut), 0, 0, L.current_stream(dev))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.fp_resize_normalize(L.ptr(frames), 256, 576, 1024, L.ptr(it), n, L.ptr(canvas), 112, 112, 4, L.ptr(lut), 0, 0, L.current_stream(dev))
    e1.record(); torch.cuda.synchronize()
    print("per_pixel" if mode else "tabled", round(e0.elapsed_time(e1) / 50 * 1e3, 1), "us")

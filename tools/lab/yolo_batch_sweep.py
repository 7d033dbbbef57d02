"""YOLOv5n-face forward at several batch sizes: does a batch whose activations fit the 256 MB memory-side cache run faster per
image than batch 256 (whose 80x80 / 40x40 tensors are 420 / 210 MB each)?    python tools/lab/yolo_batch_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from config_bench import timeit  # noqa: E402
from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd.modules.yolov5_face import preprocess_batch  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "yolov5n"
det = W.build_yolo_detector(dev, W.make_frames(16, dev, seed=6), name, cand_per_frame=80)
m = det.net
for B in (16, 32, 64, 128, 256):
    frames = W.make_frames(B, dev, seed=5)
    plan = preprocess_batch(m, frames, (640, 640))
    t = timeit(lambda: m.run_plan(plan), n=max(3, 512 // B), warm=2)
    print(f"{name} batch {B:4d}: {t * 1e3:8.3f} ms per forward = {t / B * 1e6:7.2f} us per image = {B / t:9.1f} img/s", flush=True)

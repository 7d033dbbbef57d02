"""Off-the-clock calibration of the synthetic bench workload (~64 candidates per frame before NMS and
K = 2 faces per frame after it).  Sweeps the synthetic detector's box size and prints candidates / faces per frame;
the chosen value is workload.BOX_PX."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd.pipeline import FacePipeline  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    frames = W.make_frames(128, dev, seed=1234)
    calib = W.make_frames(64, dev, seed=999)
    emb = W.build_embedder(dev)
    for box_px in [float(a) for a in sys.argv[1:]] or [90., 120., 150., 180., 210., 240.]:
        det = W.build_detector(dev, calib, box_px=box_px)
        pipe = FacePipeline(det, emb, None, max_faces_per_frame=16)
        dets, counts, _ = pipe.detect(frames)
        items, info, nf = pipe.crops(frames, dets, counts)
        cand = det.net._last_candidates[1].float().mean().item()
        it = items[:int(nf)].cpu()
        print(json.dumps({"box_px": box_px, "cand_per_frame": round(cand, 1),
                          "clusters_per_frame": round(counts.float().mean().item(), 3),
                          "faces_per_frame": round(int(nf) / frames.shape[0], 3),
                          "crop_w_mean": round(it[:, 3].float().mean().item(), 1),
                          "crop_h_mean": round(it[:, 4].float().mean().item(), 1)}), flush=True)

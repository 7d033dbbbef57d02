"""Synthetic benchmark workload (BASELINE.json configs[1]: BlazeFace back-camera, batch 256, 576x1024 frames).

No weights and no datasets exist offline, so the workload is defined here, once, for
bench.py, smoke() and the end-to-end tests:
  * frames: uint8 (B, 576, 1024, 3) BGR, flat low-contrast background plus two high-contrast textured
    patches per frame (the README's test video has two faces, /root/reference README.md:56);
  * detector: BlazeFace-back with seeded synthetic weights (synth.py); the head is conditioned so that
    random weights give well-formed boxes (w, h > 0) and scores that follow local contrast,
    and the classifier bias is calibrated once, off the clock, so that ~`cand_per_frame` anchors pass the
    score threshold per frame;
  * embedder: Mobile-FaceNet(512) with seeded synthetic weights;
  * reference set for the cosine filter: seeded unit-norm (n_ref, 512) embeddings.
Nothing here is timed; it only builds inputs.
"""
import math

import numpy as np
import torch

from .modules.blazeface.blazeface import BlazeFace, generate_anchors
from .modules.blazeface.model import BlazeFaceModel
from .modules.mobile_facenet.mobile_facenet import MobileFaceNet
from .synth import synth_state_dict

FRAME_H, FRAME_W = 576, 1024
BOX_PX = 165.0     # synthetic BlazeFace box size in model-input pixels (tools/calibrate_workload.py picks the value that
                   # gives ~2 faces per frame after the weighted NMS)


def make_frames(B, device, seed=1234, h=FRAME_H, w=FRAME_W):
    """Deterministic (seeded, generated on the host in small pieces) synthetic frames, returned on device."""
    rng = np.random.default_rng(seed)
    out = torch.empty((B, h, w, 3), dtype=torch.uint8, device=device)
    # one shared low-contrast background tile + per-frame patches keeps host generation cheap
    bg = (112 + rng.integers(-6, 7, size=(h, w, 3))).astype(np.uint8)
    bg_t = torch.from_numpy(bg).to(device)
    for i in range(B):
        f = bg_t.clone()
        for _ in range(2):
            ph = int(rng.integers(h // 4, h // 2))
            pw = ph
            y0 = int(rng.integers(0, h - ph))
            x0 = int(rng.integers(0, w - pw))
            cells = rng.integers(0, 256, size=((ph + 7) // 8, (pw + 7) // 8, 3), dtype=np.uint8)
            tex = np.repeat(np.repeat(cells, 8, axis=0), 8, axis=1)[:ph, :pw]
            f[y0:y0 + ph, x0:x0 + pw] = torch.from_numpy(tex).to(device)
        out[i] = f
    return out


def build_blazeface_back(device, seed=101, box_px=BOX_PX):
    net = BlazeFace(back_model=True)
    sd = synth_state_dict(net.state_dict(), seed, residual_gain=0.5)
    # well-formed boxes: small random offsets around the anchor, positive w/h of ~box_px (model-input pixels)
    for name in ("regressor_8", "regressor_16"):
        sd[name + ".weight"] = sd[name + ".weight"] * 0.05
        b = sd[name + ".bias"].clone() * 0.0
        b.view(-1, 16)[:, 2:4] = box_px
        sd[name + ".bias"] = b
    # scores follow feature magnitude (local contrast): non-negative classifier weights
    for name in ("classifier_8", "classifier_16"):
        sd[name + ".weight"] = sd[name + ".weight"].abs() * 0.5
        sd[name + ".bias"] = sd[name + ".bias"] * 0.0
    net.load_state_dict(sd)
    net = net.to(device)
    net.set_anchors(generate_anchors(True))
    return net


def calibrate_scores(model, frames, cand_per_frame=64):
    """Shift the classifier biases so that ~cand_per_frame anchors per frame reach min_score_thresh."""
    net = model.net
    B = frames.shape[0]
    plan = net.plan_for(B)
    from .modules.utils.image import letterbox_batch
    letterbox_batch(frames, model.input_size, net._preprocess_lut(), plan.input, pad_value=125, swap_rb=True)
    plan.run()
    c = plan.c.flatten().float()
    q = 1.0 - cand_per_frame / 896.0
    kth = torch.quantile(c[torch.randperm(c.numel(), device=c.device)[:min(c.numel(), 1_000_000)]], q)
    t = net.min_score_thresh
    delta = math.log(t / (1.0 - t)) - float(kth) + 1e-3
    with torch.no_grad():
        net.classifier_8.bias += delta
        net.classifier_16.bias += delta
    net._plans.clear()
    return delta


def build_detector(device, calib_frames, cand_per_frame=64, det_thres=0.70, bbox_area_thres=0.12, box_px=BOX_PX):
    net = build_blazeface_back(device, box_px=box_px)
    model = BlazeFaceModel("", det_thres, bbox_area_thres, "back", device=str(device), net=net)
    calibrate_scores(model, calib_frames, cand_per_frame)
    return model


def build_yolo_detector(device, calib_frames, name="yolov5s", cand_per_frame=80, det_thres=0.4, bbox_area_thres=0.12,
                        input_size=(640, 640), seed=11, box_gain=2.0, obj_spread=2.0):
    """YOLOV5FaceModel with seeded synthetic weights for BASELINE configs [2] / [3].  Random weights give arbitrary
    objectness, so -- as for BlazeFace -- the Detect biases are conditioned once, off the clock: the class
    score saturates (nc = 1), boxes are a few anchors wide (``box_gain`` raises the raw w/h logits), and the objectness
    bias is shifted so that ~``cand_per_frame`` of the 25 200 rows pass conf 0.4 per frame before NMS."""
    from .modules.yolov5_face import inference_pytorch_model_yolov5_face, preprocess_batch
    from .modules.yolov5_face.model import YOLOV5FaceModel
    from .modules.yolov5_face.yolo import Model
    m = Model(name)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed))
    m = m.fuse().to(device)
    det = m.model[-1]
    with torch.no_grad():
        for conv in det.m:
            conv.bias.view(det.na, det.no)[:, 15] += 8.0
            conv.bias.view(det.na, det.no)[:, 2:4] += box_gain
    m._plans.clear()
    in_w, in_h = input_size

    def obj_logits():
        """(B, rows) raw objectness logits in z's row order (level, anchor, y, x), read from the raw head tensors (the
        decoded z holds sigmoid(logit), which saturates at +-16 in fp32)."""
        m._plans.clear()
        plan = preprocess_batch(m, calib_frames, input_size)
        m.run_plan(plan)
        parts = []
        for h in plan.heads:                                   # (B, ny, nx, na * no)
            b, ny, nx, _ = h.shape
            parts.append(h.view(b, ny, nx, det.na, det.no)[..., 4].permute(0, 3, 1, 2).reshape(b, -1))
        return torch.cat(parts, 1).clone()

    # random weights leave the objectness logit almost constant over the image (std ~0.01) and offset per (level,
    # anchor) by the random bias: normalise every (level, anchor) group to the same mean and a spread of ~obj_spread so
    # that candidates come from every level / anchor and follow the image content ...
    logit = obj_logits()
    row = 0
    with torch.no_grad():
        for lvl, conv in enumerate(det.m):
            npos = (in_h // int(det.stride[lvl])) * (in_w // int(det.stride[lvl]))
            for a in range(det.na):
                grp = logit[:, row:row + npos]
                mu, sd = float(grp.mean()), float(grp.std())
                g = obj_spread / max(sd, 1e-6)
                conv.weight.view(det.na, det.no, -1)[a, 4] *= g
                b = conv.bias.view(det.na, det.no)
                b[a, 4] = g * (b[a, 4] - mu)                   # group mean -> 0
                row += npos
    # ... then shift it so that ~cand_per_frame rows per frame pass conf 0.4
    logit = obj_logits().flatten()
    k = min(int(cand_per_frame * calib_frames.shape[0]), logit.numel() - 1)
    kth = torch.topk(logit, k + 1).values[-1]
    delta = math.log(0.4 / 0.6) - float(kth) + 1e-3
    with torch.no_grad():
        for conv in det.m:
            conv.bias.view(det.na, det.no)[:, 4] += delta
    m._plans.clear()
    return YOLOV5FaceModel(m, det_thres, bbox_area_thres, inference_pytorch_model_yolov5_face, input_size)


def build_embedder(device, seed=300):
    net = MobileFaceNet(512)
    net.load_state_dict(synth_state_dict(net.state_dict(), seed))
    return net.to(device)


def make_reference(n_ref, device, seed=43, dim=512):
    rng = np.random.default_rng(seed)
    r = rng.normal(0, 1, (n_ref, dim)).astype(np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    return torch.from_numpy(r).to(device)

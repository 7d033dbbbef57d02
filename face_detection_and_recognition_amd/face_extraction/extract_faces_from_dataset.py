"""Batched detect -> crop -> embed for media items, with the reference's on-disk feature format
(face_detection_and_extraction/face_extraction/extract_faces_from_dataset.py:270-365).

  extract_face_feat_conf_area_list(pipe, frames)  :270-307  (here: a whole batch of frames per call, on device)
  save_extracted_faces(...)                       :311-363  (same annot dict, zero-padded feature vector, np.save)

The reference walks a dataset one frame at a time through a Net wrapper; this module is the SURVEY 8(f) rank-1
"next" row: the same composition driven by pipeline.FacePipeline, so detection, cropping and embedding of all frames
of a media item are three device-resident stages.  Directory walking / video decoding stay with the caller."""
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

MAX_N_FACES_PER_FRAME = 3      # extract_faces_from_dataset.py:38
MAX_N_FRAME_FROM_VID = 15      # :40


@dataclass
class FrameFacesObj:
    """One frame's faces (the reference's FrameFacesObj record)."""
    frame_num: int
    time_sec: float
    confs: List[float]
    areas: List[float]
    boxes: np.ndarray
    feats: List[np.ndarray] = field(default_factory=list)


def extract_face_feat_conf_area_list(pipe, frames, frame_nums=None, times_sec=None) -> List[FrameFacesObj]:
    """frames: (B, H, W, 3) u8 BGR (numpy or CUDA tensor).  One FacePipeline step (no similarity filter needed);
    returns per-frame records with boxes (orig pixels, rounded), confs, area fractions and embeddings."""
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(np.ascontiguousarray(frames))
    frames = frames.to(pipe.dev)
    B, H, W, _ = frames.shape
    res = pipe.step(frames)                   # detect -> crops -> embed (+ the exact re-run on a detector overflow)
    n = res["n_faces"]
    emb = res["emb"].cpu().numpy()
    info = res["info"].cpu().numpy()
    out = [FrameFacesObj(frame_nums[i] if frame_nums is not None else i,
                         times_sec[i] if times_sec is not None else 0.0, [], [], np.zeros((0, 4), np.float32))
           for i in range(B)]
    boxes = [[] for _ in range(B)]
    for k in range(n):
        f = int(info[k, 0])
        x1, y1, x2, y2, conf, area = info[k, 1:7]
        boxes[f].append([x1, y1, x2, y2])
        out[f].confs.append(float(conf))
        out[f].areas.append(float(area))     # BlazeFace: fraction (inference.py:40-46); YOLO: percent (onnx_utils.py:331)
        out[f].feats.append(emb[k])
    for f in range(B):
        if boxes[f]:
            out[f].boxes = np.asarray(boxes[f], dtype=np.float32)
    return out


def save_extracted_faces(frames_faces_obj_list, media_root, class_name, feats_save_dir, face_feature_size,
                         class2label_dict, save_feat=True):
    """:311-363 without the cv2.imwrite branch: annot dict {media_id, frames_info, class_name, label, feature}
    with the feature vector zero-padded to MAX_N_FRAME_FROM_VID * MAX_N_FACES_PER_FRAME * face_feature_size."""
    annot = {"media_id": media_root, "frames_info": []}
    feats_list, total = [], 0
    for fr in frames_faces_obj_list:
        if save_feat:
            feats = list(fr.feats[:MAX_N_FACES_PER_FRAME])
            feats.extend([np.zeros(face_feature_size)] * (MAX_N_FACES_PER_FRAME - len(feats)))
            feats_list.extend(feats)
        annot["frames_info"].append({"frame_num": fr.frame_num, "time_sec": fr.time_sec, "confs": fr.confs,
                                     "areas": fr.areas})
        total += len(fr.confs)
    os.makedirs(feats_save_dir, exist_ok=True)
    annot["class_name"] = class_name
    annot["label"] = class2label_dict[class_name]
    if save_feat:
        if len(frames_faces_obj_list) < MAX_N_FRAME_FROM_VID:
            pad = MAX_N_FRAME_FROM_VID - len(frames_faces_obj_list)
            feats_list.extend([np.zeros(face_feature_size)] * (MAX_N_FACES_PER_FRAME * pad))
        annot["feature"] = np.concatenate(feats_list, axis=0).astype(np.float32)
    np.save(os.path.join(feats_save_dir, media_root + ".npy"), annot)
    return total

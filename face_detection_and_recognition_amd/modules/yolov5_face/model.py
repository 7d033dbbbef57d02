"""YOLOV5FaceModel plugin (face_detection_and_extraction/modules/yolov5_face/model.py:8-37) + batched path."""
from typing import Any, Callable, Tuple

import numpy as np
import torch

from ..models.base import Model
from . import nms_face_device, preprocess_batch
from .general import MAX_DET


class YOLOV5FaceModel(Model):

    __slots__ = ["net", "inf_func"]
    dets_fmt = 1     # rows are (x1, y1, x2, y2, conf, ...) in model-input pixels (pipeline.py / fp_dets_to_crops)

    def __init__(self, net: Any, det_thres: float, bbox_area_thres: float, inf_func: Callable,
                 input_size: Tuple[int, int]):
        Model.__init__(self, input_size, det_thres, bbox_area_thres)
        self.net = net
        self.inf_func = inf_func

    def __call__(self, cv2_img: np.ndarray) -> np.ndarray:
        """model.py:23-37: (k, 5) [xmin, ymin, xmax, ymax, conf] normalised to [0, 1]."""
        iw, ih = self.input_size
        detections = self.inf_func(self.net, cv2_img, self.input_size)
        if detections is not None and len(detections):
            detections = detections.cpu().numpy()
            detections[:, :4] = detections[:, :4] / np.array([iw, ih, iw, ih])
            detections = detections[:, :5]
        else:
            detections = np.zeros((0, 5), dtype=np.float32)   # the reference returns an uninitialised (0, 5) array
        return detections

    def raw_batch(self, frames, max_det=MAX_DET):
        """frames (B, H, W, 3) u8 BGR -> device dets (B, max_det, 16) in input pixels, counts (B,), overflow (B,).
        overflow[i] != 0: image i had more than max_det survivors and was truncated; the caller re-runs with
        ``max_det=None`` (no cap, as non_max_suppression_face does) -- FacePipeline.step reads the flag in the same
        host transfer as the face count."""
        dev = self.net._device()
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        frames = frames.to(dev)
        plan = preprocess_batch(self.net, frames, self.input_size)
        z = self.net.run_plan(plan)
        out, cnt, _, over = nms_face_device(z, conf_thres=0.4, iou_thres=0.5,
                                            max_det=z.shape[1] if max_det is None else max_det)
        return out, cnt, over

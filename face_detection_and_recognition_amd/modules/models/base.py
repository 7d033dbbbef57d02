"""Plugin API of the reference, unchanged in shape
(face_detection_and_extraction/modules/models/base.py:6-58)."""
from typing import Any, List, Optional, Tuple

import numpy as np


class Model:
    """Base detector: ``__call__(cv2_img BGR HWC u8) -> np.ndarray[N, 4(+lmk)+1]`` with
    coordinates normalised to [0, 1] w.r.t. ``input_size`` (base.py:6-31)."""

    __slots__ = ["input_size", "det_thres", "bbox_area_thres", "returns_opt_labels"]

    def __init__(self, input_size: Tuple[int, int], det_thres: float, bbox_area_thres: float,
                 returns_opt_labels: bool = False):
        self.input_size = input_size          # (width, height)
        self.det_thres = det_thres
        self.bbox_area_thres = bbox_area_thres
        self.returns_opt_labels = returns_opt_labels

    def __call__(self):
        raise NotImplementedError("__call__ method has not been implemented")


class PostProcessedDetection:
    """Post-processed detections for one image (base.py:34-58)."""

    __slots__ = ["boxes", "bbox_confs", "bbox_areas", "bbox_lmarks", "bbox_labels"]

    def __init__(self, boxes: np.ndarray, bbox_confs: np.ndarray, bbox_areas: np.ndarray,
                 bbox_lmarks: Optional[np.ndarray] = None, bbox_labels: Optional[List[Any]] = None):
        self.boxes = boxes
        self.bbox_confs = bbox_confs
        self.bbox_areas = bbox_areas
        self.bbox_lmarks = bbox_lmarks
        self.bbox_labels = bbox_labels

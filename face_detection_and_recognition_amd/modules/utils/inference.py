"""Inference drivers (face_detection_and_extraction/modules/utils/inference.py).  The GUI loops
(cv2.imshow / VideoCapture) are out of scope; ``inference_img`` keeps its signature and returns the
post-processed detections instead of drawing them."""
import os
from typing import Any, List, Optional, Tuple

import numpy as np

from ..models.base import Model, PostProcessedDetection
from .image import scale_coords


def get_dets_bboxes_confs_lmarks_areas(dets: np.ndarray, orig_size: Tuple[int, int], in_size: Tuple[int, int],
                                       det_thres: float, bbox_area_thres: float,
                                       opt_labels: Optional[List[Any]] = None) -> PostProcessedDetection:
    """inference.py:11-58: threshold (strict >), de-normalise to the model input size, area filter
    (100*fraction > thres; the stored value is the fraction), un-letterbox, round."""
    w, h = orig_size
    iw, ih = in_size
    dets = dets[dets[:, -1] > det_thres]
    dets[:, :-1] = dets[:, :-1] * np.array([iw, ih] * ((dets.shape[-1] - 1) // 2))
    bbox_area_perc = ((dets[:, 2] - dets[:, 0]) * (dets[:, 3] - dets[:, 1])) / (iw * ih)
    keep = (100 * bbox_area_perc) > bbox_area_thres
    dets, bbox_area_perc = dets[keep], bbox_area_perc[keep]
    confs = dets[:, -1]
    dets = scale_coords((ih, iw), dets[:, :-1], (h, w)).round()
    return PostProcessedDetection(boxes=dets[:, :4], bbox_confs=confs, bbox_areas=bbox_area_perc,
                                  bbox_lmarks=dets[:, 4:], bbox_labels=opt_labels)


def load_image(path: str, device=None):
    """BGR HWC u8 like cv2.imread (inference.py:68-76).  device = a HIP device: the frame is decoded there (baseline JPEGs:
    host Huffman + device reconstruction, modules/utils/jpeg.py, byte-identical to libjpeg-turbo) and returned as a device
    tensor; device = None: numpy array decoded by Pillow on the host (cv2 is not a dependency of this build)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} does not exist")
    if device is not None:
        from .jpeg import imread
        return imread(path, device, bgr=True)
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])


def inference_img(net: Model, img, waitKey_val: int = 0) -> PostProcessedDetection:
    """inference.py:61-93 without the drawing/imshow tail."""
    # a net that takes device frames gets the file decoded ON its device (baseline JPEG: host Huffman + device IDCT / upsampling
    # / colour conversion, byte-identical to cv2.imread's libjpeg-turbo); everything else the host array, as the reference
    dev = net.net._device() if getattr(net, "accepts_device_frames", False) else None
    image = load_image(img, dev if dev is not None and dev.type == "cuda" else None) if isinstance(img, str) else img
    h, w = image.shape[:2]
    dets = net(image)
    labels = None
    if net.returns_opt_labels:
        dets, labels = dets
    return get_dets_bboxes_confs_lmarks_areas(dets, (w, h), net.input_size, net.det_thres, net.bbox_area_thres, labels)

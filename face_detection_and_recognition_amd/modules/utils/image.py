"""Image / coordinate helpers with the reference's names
(face_detection_and_extraction/modules/utils/image.py).  The pixel work (resize, pad, normalise) runs in
fp_resize_normalize (csrc/image.hip); the per-detection coordinate helpers are the same small numpy host
arithmetic as the reference (a handful of boxes per image, fp64).  Drawing helpers are out of scope."""
import math
from typing import Tuple

import numpy as np
import torch

from ... import _lib as L


def make_divisible(x, divisor):
    """image.py:15-19."""
    return math.ceil(x / divisor) * divisor


def check_img_size(img_size: int, s: int = 32):
    """image.py:22-31: round img_size up to a multiple of the max stride."""
    new_size = make_divisible(img_size, int(s))
    if new_size != img_size:
        print('WARNING: --img-size %g must be multiple of max stride %g, updating to %g' % (img_size, s, new_size))
    return new_size


def letterbox_geometry(in_w, in_h, new_w, new_h):
    """Geometry of pad_resize_image (image.py:42-54): resized size (int() truncation) and left/top pad."""
    scale = min(new_w / in_w, new_h / in_h)
    sw, sh = int(in_w * scale), int(in_h * scale)
    d_w, d_h = max(new_w - sw, 0), max(new_h - sh, 0)
    return sw, sh, d_w // 2, d_h // 2


def letterbox_batch(frames_u8, new_size, lut, canvas, pad_value=125, swap_rb=False):
    """Batched pad_resize_image + normalisation on device.
    frames_u8: (B, H, W, 3) u8 CUDA tensor; canvas: (B, new_h, new_w, C>=3) fp32 CUDA tensor (written);
    lut: (256,) fp32 CUDA tensor (normalisation of each u8 value)."""
    lib = L.load()
    B, H, W, _ = frames_u8.shape
    new_w, new_h = new_size
    assert canvas.shape[0] == B and canvas.shape[1] == new_h and canvas.shape[2] == new_w
    sw, sh, left, top = letterbox_geometry(W, H, new_w, new_h)
    items = torch.tensor([[i, 0, 0, W, H, left, top, sw, sh] for i in range(B)], dtype=torch.int32,
                         device=frames_u8.device)
    L.check(lib.fp_resize_normalize(L.ptr(frames_u8), B, H, W, L.ptr(items), B, L.ptr(canvas), new_h, new_w,
                                    canvas.shape[3], L.ptr(lut), int(pad_value), int(bool(swap_rb)),
                                    L.current_stream(frames_u8.device)), "fp_resize_normalize")
    return canvas


def bind_letterbox(plan, frames_u8, lut, pad_value=125, swap_rb=False):
    """Point a plan whose first op reads u8 frames (FP_OP_*_U8) at a batch: external buffers [frames, tap tables, LUT].
    The tap tables of the plan's (frame size -> canvas size) letterbox are computed once per plan on the device
    (fp_letterbox_tables: the same arithmetic as fp_resize_normalize, one entry per canvas column / row)."""
    B, fh, fw, _ = frames_u8.shape
    assert plan.frame_hw == (fh, fw) and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
    ch, cw = plan.canvas_hw
    key = (int(pad_value), bool(swap_rb))
    if plan.tables is None or plan.tables[0] != key:
        sw, sh, left, top = letterbox_geometry(fw, fh, cw, ch)
        t = torch.empty(((cw + ch + 2) * 2,), dtype=torch.int32, device=frames_u8.device)
        L.check(L.load().fp_letterbox_tables(fh, fw, ch, cw, 0, 0, fw, fh, left, top, sw, sh, int(pad_value),
                                             int(bool(swap_rb)), L.ptr(t), L.current_stream(frames_u8.device)),
                "fp_letterbox_tables")
        plan.tables = (key, t)
    plan.set_ext([frames_u8, plan.tables[1], lut])
    return plan


def pad_resize_image(cv2_img: np.ndarray, new_size: Tuple[int, int] = (640, 480),
                     color: Tuple[int, int, int] = (125, 125, 125), device="cuda") -> np.ndarray:
    """image.py:31-59 for one image (numpy in, numpy out) through the device kernel."""
    if not (color[0] == color[1] == color[2]):
        raise NotImplementedError("pad colour must be grey (the reference always uses (125,125,125))")
    new_w, new_h = new_size
    in_h, in_w = cv2_img.shape[:2]
    sw, sh, _, _ = letterbox_geometry(in_w, in_h, new_w, new_h)
    frames = torch.from_numpy(np.ascontiguousarray(cv2_img)).to(device).unsqueeze(0)
    out_h, out_w = max(new_h, sh), max(new_w, sw)
    canvas = torch.empty((1, out_h, out_w, 3), dtype=torch.float32, device=device)
    lut = torch.arange(256, dtype=torch.float32, device=device)
    letterbox_batch(frames, (out_w, out_h), lut, canvas, pad_value=color[0])
    return canvas[0].to(torch.uint8).cpu().numpy()


def clip_coords(boxes, img_shape: Tuple[int, int]):
    """image.py:62-76: clip xyxy (first four columns) to (height, width), in place."""
    if boxes.any():
        if isinstance(boxes, np.ndarray):
            for col, lim in ((0, img_shape[1]), (1, img_shape[0]), (2, img_shape[1]), (3, img_shape[0])):
                np.clip(boxes[:, col], 0, lim, out=boxes[:, col])
        else:
            for col, lim in ((0, img_shape[1]), (1, img_shape[0]), (2, img_shape[1]), (3, img_shape[0])):
                boxes[:, col].clamp_(0, lim)


def scale_coords(img1_shape: Tuple[int, int], coords: np.ndarray, img0_shape: Tuple[int, int], ratio_pad=None):
    """image.py:79-99: undo the letterbox (subtract pad, divide by gain, clip the box columns)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    ncol = coords.shape[-1]
    coords[:, [i for i in range(ncol) if i % 2 == 0]] -= pad[0]
    coords[:, [i for i in range(ncol) if i % 2 == 1]] -= pad[1]
    coords /= gain
    clip_coords(coords, img0_shape)
    return coords


def standardize_image(cv2_img: np.ndarray, new_dtype=np.float32):
    """image.py:102-118 (== tf.image.per_image_standardization used by filter_faces_using_reference.py:67)."""
    if cv2_img.ndim == 4:
        axis, size = (1, 2, 3), cv2_img[0].size
    elif cv2_img.ndim == 3:
        axis, size = (0, 1, 2), cv2_img.size
    else:
        raise ValueError('Dimension should be 3 or 4')
    mean = np.mean(cv2_img, axis=axis, keepdims=True)
    std = np.std(cv2_img, axis=axis, keepdims=True)
    return ((cv2_img - mean) / np.maximum(std, 1.0 / np.sqrt(size))).astype(new_dtype)


def calculate_bbox_iou(bbox1, bbox2):
    """image.py:124-143."""
    x_diff = min(bbox1[2], bbox2[2]) - max(bbox1[0], bbox2[0])
    y_diff = min(bbox1[3], bbox2[3]) - max(bbox1[1], bbox2[1])
    if x_diff < 0 or y_diff < 0:
        return 0
    inter = x_diff * y_diff
    return inter / ((bbox1[2] - bbox1[0]) * (bbox1[3] - bbox1[1]) + (bbox2[2] - bbox2[0]) * (bbox2[3] - bbox2[1]) - inter)

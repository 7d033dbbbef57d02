"""Oracle for the image front end and the box post-processing.  TEST INFRASTRUCTURE ONLY.

resize_bilinear_u8 restates OpenCV's INTER_LINEAR path for 8-bit images (opencv-python 4.11.0.86 is the
reference's pin, pyproject.toml; the package and its sources are absent here, so this is restated from
the published algorithm in modules/imgproc/src/resize.cpp: 11-bit fixed-point coefficients,
HResizeLinear + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>).  PARITY UNPINNED vs cv2 itself.
"""
import math

import numpy as np


def _coefs(dsize, ssize):
    scale = float(ssize) / float(dsize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= ssize - 1
    f[hi], s[hi] = 0.0, ssize - 1
    s1 = np.minimum(s + 1, ssize - 1)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, s1, a0, a1


def resize_bilinear_u8(img, dsize):
    """cv2.resize(img, (dw, dh)) for HxWxC uint8, INTER_LINEAR."""
    dw, dh = dsize
    img = np.asarray(img)
    sh, sw = img.shape[:2]
    sx0, sx1, ax0, ax1 = _coefs(dw, sw)
    sy0, sy1, by0, by1 = _coefs(dh, sh)
    src = img.astype(np.int64)
    hrow = src[:, sx0] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]      # (sh, dw, C), x2048
    h0, h1 = hrow[sy0], hrow[sy1]
    out = (((by0[:, None, None] * (h0 >> 4)) >> 16) + ((by1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(in_w, in_h, new_w, new_h):
    """pad_resize_image geometry (fde/modules/utils/image.py:42-54)."""
    scale = min(new_w / in_w, new_h / in_h)
    sw, sh = int(in_w * scale), int(in_h * scale)
    d_w, d_h = max(new_w - sw, 0), max(new_h - sh, 0)
    return sw, sh, d_w // 2, d_h // 2


def pad_resize_image(img, new_size=(640, 480), color=(125, 125, 125)):
    """pad_resize_image (fde/modules/utils/image.py:31-59)."""
    in_h, in_w = img.shape[:2]
    new_w, new_h = new_size
    sw, sh, left, top = letterbox_geometry(in_w, in_h, new_w, new_h)
    res = resize_bilinear_u8(img, (sw, sh))
    out = np.empty((max(new_h, sh), max(new_w, sw), 3), dtype=np.uint8)
    out[:] = np.asarray(color, dtype=np.uint8)
    out[top:top + sh, left:left + sw] = res
    return out


def clip_coords(boxes, img_shape):
    """clip_coords (image.py:62-76): first four columns only."""
    if boxes.any():
        boxes[:, 0].clip(0, img_shape[1], out=boxes[:, 0])
        boxes[:, 1].clip(0, img_shape[0], out=boxes[:, 1])
        boxes[:, 2].clip(0, img_shape[1], out=boxes[:, 2])
        boxes[:, 3].clip(0, img_shape[0], out=boxes[:, 3])


def scale_coords(img1_shape, coords, img0_shape):
    """scale_coords (image.py:79-99), ratio_pad=None."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    xs = [i for i in range(coords.shape[-1]) if i % 2 == 0]
    ys = [i for i in range(coords.shape[-1]) if i % 2 == 1]
    coords[:, xs] -= pad[0]
    coords[:, ys] -= pad[1]
    coords /= gain
    clip_coords(coords, img0_shape)
    return coords


def dets_to_boxes(dets, orig_size, in_size, det_thres, bbox_area_thres):
    """get_dets_bboxes_confs_lmarks_areas (fde/modules/utils/inference.py:11-58).
    Returns dict(boxes, bbox_confs, bbox_areas, bbox_lmarks)."""
    w, h = orig_size
    iw, ih = in_size
    dets = dets[dets[:, -1] > det_thres]
    dets[:, :-1] = dets[:, :-1] * np.array([iw, ih] * ((dets.shape[-1] - 1) // 2))
    area = (dets[:, 2] - dets[:, 0]) * (dets[:, 3] - dets[:, 1])
    perc = area / (iw * ih)
    keep = (100 * perc) > bbox_area_thres
    dets, perc = dets[keep], perc[keep]
    confs = dets[:, -1]
    dets = scale_coords((ih, iw), dets[:, :-1], (h, w)).round()
    return dict(boxes=dets[:, :4], bbox_confs=confs, bbox_areas=perc, bbox_lmarks=dets[:, 4:])


def standardize_image(img, new_dtype=np.float32):
    """standardize_image (image.py:102-118) == tf.image.per_image_standardization
    (similar_face_filtering/filter_faces_using_reference.py:67)."""
    img = np.asarray(img)
    axis = (1, 2, 3) if img.ndim == 4 else (0, 1, 2)
    size = img[0].size if img.ndim == 4 else img.size
    mean = np.mean(img, axis=axis, keepdims=True)
    std = np.std(img, axis=axis, keepdims=True)
    return ((img - mean) / np.maximum(std, 1.0 / np.sqrt(size))).astype(new_dtype)


def blaze_lut():
    """x.float()/127.5 - 1.0 (blazeface.py:248-250) for the 256 u8 values, torch fp32 semantics."""
    import torch
    return (torch.arange(256, dtype=torch.float32) / 127.5 - 1.0).numpy()


def yolo_lut():
    """img.astype(float32) / 255.0 (y5/__init__.py:18-19)."""
    x = np.arange(256).astype(np.float32)
    x /= 255.0
    return x


def mfn_lut():
    """(resized - 127.5) / 127.5 in float64, then astype(float32) (mobile_facenet/utils.py:13-16)."""
    return ((np.arange(256, dtype=np.uint8) - 127.5) / 127.5).astype(np.float32)


def crop_face(frame, box, offsets=(-6, -1, 4, 5)):
    """extract_faces_from_dataset.py:289-303: int(), offsets (tx,ty,bx,by), clamp, slice."""
    h, w = frame.shape[:2]
    tx, ty, bx, by = offsets
    x, y, xw, yh = (int(v) for v in box)
    x, y, xw, yh = max(x + tx, 0), max(y + ty, 0), min(xw + bx, w), min(yh + by, h)
    return frame[y:yh, x:xw], (x, y, xw, yh)


def tf_resize_bilinear(img_f32, size):
    """tf.image.resize(img, size) for TF2 (bilinear, half_pixel_centers=True, antialias=False): img (H, W, C) fp32,
    size (oh, ow).  Restated from TF's resize_bilinear_op (compute_interpolation_weights + lerp order): PARITY UNPINNED
    (TensorFlow is absent offline)."""
    H, W = img_f32.shape[:2]
    oh, ow = size

    def weights(o, i):
        scale = np.float32(i) / np.float32(o)
        f = (np.arange(o, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
        fl = np.floor(f)
        lo = np.maximum(fl.astype(np.int64), 0)
        hi = np.minimum(np.ceil(f).astype(np.int64), i - 1)
        return lo, hi, (f - fl).astype(np.float32)

    y0, y1, ly = weights(oh, H)
    x0, x1, lx = weights(ow, W)
    img = img_f32.astype(np.float32)
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = tl + (tr - tl) * lx[None, :, None]
    bot = bl + (br - bl) * lx[None, :, None]
    return top + (bot - top) * ly[:, None, None]


def read_and_preprocess_rgb(img_u8_rgb, in_size=(160, 160)):
    """similar_face_filtering/filter_faces_using_reference.py:60-68 after the JPEG decode: convert_image_dtype ->
    resize -> per_image_standardization (formula pinned by sff/tests/base/test_similar_faces_filter.py:19-27)."""
    x = img_u8_rgb.astype(np.float32) * np.float32(1.0 / 255.0)
    x = tf_resize_bilinear(x, in_size)
    return standardize_image(x)

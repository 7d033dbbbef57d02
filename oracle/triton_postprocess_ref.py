"""ORACLE (test infrastructure only): CPU restatement of the Triton python post-process model
(fde/modules/face_detection_trt_server/models/yolov5_face_postprocess/1/model.py:32-113 and its utils.py).
Decode and w_non_max_suppression reuse oracle/yolo_ref.py (pinned by tests/golden/yolo_decode_wnms.npz, produced by
the reference's own onnx_utils functions, of which the server's utils.py is a verbatim twin).  cv2.resize on float
data is restated (cv2 is absent offline): PARITY UNPINNED for the resize, as for the u8 path in image_ref.py.
"""
import numpy as np
import torch

from . import yolo_ref

ANCHORS = ((4., 5., 8., 10., 13., 16.), (23., 29., 43., 55., 73., 105.), (146., 217., 231., 300., 335., 433.))


def resize_bilinear_f32(img, dsize):
    """cv2.resize(img_f32_hwc, (dw, dh)) with INTER_LINEAR: horizontal pass then vertical, float weights."""
    dw, dh = dsize
    sh, sw = img.shape[:2]

    def taps(d, s):
        idx = np.arange(d)
        fs = ((idx + 0.5) * (s / d) - 0.5).astype(np.float32)
        s0 = np.floor(fs).astype(np.int64)
        a = (fs - s0).astype(np.float32)
        lo = s0 < 0
        s0[lo], a[lo] = 0, 0
        hi = s0 >= s - 1
        s0[hi], a[hi] = s - 1, 0
        return s0, np.minimum(s0 + 1, s - 1), a

    x0, x1, ax = taps(dw, sw)
    y0, y1, ay = taps(dh, sh)
    img = img.astype(np.float32)
    hz = img[:, x0] * (np.float32(1) - ax)[None, :, None] + img[:, x1] * ax[None, :, None]
    return hz[y0] * (np.float32(1) - ay)[:, None, None] + hz[y1] * ay[:, None, None]


def execute(stride_8, stride_16, stride_32, images, face_det_thres, face_bbox_area_thres, out_size=(112, 112)):
    """model.py:32-113 for one request -> (faces, bboxes, confs) numpy arrays."""
    input_image = np.transpose(np.asarray(images, np.float32)[0], (1, 2, 0)) * np.float32(255.0)
    heads = [torch.from_numpy(np.asarray(s, np.float32)) for s in (stride_8, stride_16, stride_32)]
    outputx = _decode(heads)
    detections = yolo_ref.w_non_max_suppression(outputx, conf_thres=0.4, nms_thres=0.3)[0]
    mow, moh = out_size
    h, w = input_image.shape[:2]
    if detections is None:
        return (np.zeros((1, 3, moh, mow), np.float32), np.asarray([[0, 0, 0, 0]], np.int32),
                np.asarray([[0.]], np.float32))
    input_image = input_image[..., ::-1]
    det = detections.numpy()
    det = det[det[..., 4] > face_det_thres]
    bbox_area = (det[:, 2] - det[:, 0]) * (det[:, 3] - det[:, 1])
    det = det[100 * bbox_area / (w * h) > face_bbox_area_thres]
    faces, boxes = [], []
    for box in det[..., :4]:
        xmin, ymin, xmax, ymax = map(int, box)
        x, y, xw, yh = max(xmin, 0), max(ymin, 0), min(xmax, w), min(ymax, h)
        face = resize_bilinear_f32(input_image[y:yh, x:xw].copy(), (mow, moh)).astype(np.float32)
        face = (face - np.float32(127.5)) / np.float32(127.5)
        faces.append(np.transpose(face, (2, 0, 1)))
        boxes.append(np.asarray([x, y, xw, yh], np.int32))
    return np.asarray(faces, np.float32), np.asarray(boxes), np.asarray(det[..., 4], np.float32)


def _decode(heads):
    """utils.py:11-52 conv_strides_to_anchors: heads (bs, 3, ny, nx, 16) -> (bs, n, 16)."""
    z = []
    for x, stride, anc in zip(heads, (8., 16., 32.), ANCHORS):
        bs, na, ny, nx, no = x.shape
        yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
        ag = torch.tensor(anc).view(1, na, 1, 1, 2)
        y = torch.zeros_like(x)
        y[..., [0, 1, 2, 3, 4, 15]] = x[..., [0, 1, 2, 3, 4, 15]].sigmoid()
        y[..., 5:15] = x[..., 5:15]
        y[..., 0:2] = (y[..., 0:2] * 2. - 0.5 + grid) * stride
        y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * ag
        for k in range(5, 15, 2):
            y[..., k:k + 2] = y[..., k:k + 2] * ag + grid * stride
        z.append(y.view(bs, -1, no))
    return torch.cat(z, 1)

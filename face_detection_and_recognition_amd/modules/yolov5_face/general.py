"""Post-processing API of the YOLOv5-face path with the reference's names:
  non_max_suppression_face  (face_detection_and_extraction/modules/yolov5_face/pytorch/utils/general.py:370-453)
  w_non_max_suppression     (face_detection_and_extraction/modules/yolov5_face/onnx/onnx_utils.py:107-163)
  conv_strides_to_anchors   (onnx_utils.py:30-73)
  get_bboxes_confs_areas    (onnx_utils.py:313-340)
All box arithmetic runs in csrc/post.hip, one workgroup per image."""
import numpy as np
import torch

from ... import _lib as L
from ..utils.image import scale_coords

MAX_DET = 2048       # kept boxes per image of the fast path; beyond it the call is repeated without a cap


def _nms_common(fn_name, prediction, conf_thres, iou_thres, max_det, out_cols):
    lib = L.load()
    assert prediction.is_cuda and prediction.dtype == torch.float32 and prediction.shape[-1] == 16
    pred = prediction.contiguous()
    B, n_rows, _ = pred.shape
    dev = pred.device
    out = torch.empty((B, max_det, out_cols), dtype=torch.float32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    keep = torch.empty((B, max_det), dtype=torch.int32, device=dev)
    over = torch.empty((B,), dtype=torch.int32, device=dev)
    max_cand = (n_rows + 3) // 4 * 4          # every row may be a candidate, as in the reference
    nbytes = lib.fp_yolo_nms_scratch_bytes(B, max_cand)
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    fn = getattr(lib, fn_name)
    L.check(fn(L.ptr(pred), B, n_rows, float(conf_thres), float(iou_thres), max_cand, max_det, L.ptr(out), L.ptr(cnt),
               L.ptr(keep), L.ptr(over), L.ptr(scratch), nbytes, L.current_stream(dev)), fn_name)
    return out, cnt, keep, over


def nms_face_device(prediction, conf_thres=0.25, iou_thres=0.45, max_det=MAX_DET):
    """Device-resident form: (out [B, max_det, 16], counts [B], keep_idx [B, max_det], overflow [B])."""
    return _nms_common("fp_yolo_nms", prediction, conf_thres, iou_thres, max_det, 16)


def non_max_suppression_face(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, labels=()):
    """general.py:370-453 for nc = 1: list (one per image) of (k, 16) tensors
    (x1, y1, x2, y2, conf, 10 landmark coords, cls), kept boxes in score order."""
    if classes is not None or labels:
        raise NotImplementedError("class filtering / autolabelling are not part of the face path (nc = 1)")
    out, cnt, _, over = nms_face_device(prediction, conf_thres, iou_thres)
    if int(over.sum()):      # more than MAX_DET survivors in some image (untrained weights): exact re-run, no cap
        out, cnt, _, over = nms_face_device(prediction, conf_thres, iou_thres, max_det=prediction.shape[1])
    cnt = cnt.cpu().tolist()
    return [out[i, :k].clone() if k > 0 else torch.zeros((0, 16), device=prediction.device)
            for i, k in enumerate(cnt)]


def w_non_max_suppression(prediction, num_classes=1, conf_thres=0.5, nms_thres=0.4):
    """onnx_utils.py:107-163 for num_classes = 1: list of (k, 7) tensors or None per image."""
    assert num_classes == 1
    out, cnt, _, over = _nms_common("fp_yolo_w_nms", prediction, conf_thres, nms_thres, MAX_DET, 7)
    if int(over.sum()):
        out, cnt, _, over = _nms_common("fp_yolo_w_nms", prediction, conf_thres, nms_thres, prediction.shape[1], 7)
    cnt = cnt.cpu().tolist()
    return [out[i, :k].clone() if k > 0 else None for i, k in enumerate(cnt)]


def conv_strides_to_anchors(pred, device="cuda"):
    """onnx_utils.py:30-73: the three raw head tensors (bs, 3, ny, nx, 16) of an exported model -> (bs, n, 16)."""
    lib = L.load()
    strides = (8.0, 16.0, 32.0)
    anchors = ((4., 5., 8., 10., 13., 16.), (23., 29., 43., 55., 73., 105.), (146., 217., 231., 300., 335., 433.))
    dev = torch.device(str(device).replace("hip", "cuda"))
    heads = [torch.as_tensor(p).to(dev, torch.float32) for p in pred]
    bs = heads[0].shape[0]
    n_rows = sum(h.shape[1] * h.shape[2] * h.shape[3] for h in heads)
    z = torch.empty((bs, n_rows, 16), dtype=torch.float32, device=dev)
    row = 0
    for h, s, a in zip(heads, strides, anchors):
        _, na, ny, nx, no = h.shape
        nhwc = h.permute(0, 2, 3, 1, 4).reshape(bs, ny, nx, na * no).contiguous()
        anc = (L.C.c_float * 6)(*a)
        L.check(lib.fp_yolo_decode(L.ptr(nhwc), bs, ny, nx, na, s, anc, L.ptr(z), n_rows * 16, row,
                                   L.current_stream(dev)), "fp_yolo_decode")
        row += na * ny * nx
    return z


def get_bboxes_confs_areas(dets, det_thres, bbox_area_thres, orig_size, in_size):
    """onnx_utils.py:313-340 (host arithmetic on the handful of kept boxes, as in the reference)."""
    w, h = orig_size
    iw, ih = in_size
    if not isinstance(dets, np.ndarray):
        dets = dets.cpu().numpy()
    dets = dets[dets[..., 4] > det_thres]
    bbox_area_perc = 100 * ((dets[:, 2] - dets[:, 0]) * (dets[:, 3] - dets[:, 1])) / (iw * ih)
    dets = dets[bbox_area_perc > bbox_area_thres]
    boxes = scale_coords((ih, iw), dets[..., :4], (h, w)).round()
    return boxes, dets[..., 4], bbox_area_perc

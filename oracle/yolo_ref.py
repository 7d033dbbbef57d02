"""Oracle for the YOLOv5-face path (y5 = fde/modules/yolov5_face/pytorch).  TEST INFRASTRUCTURE ONLY.

forward() restates Model.forward_once (y5/models/yolo.py:177-198) over the yaml layer table with torch-CPU
functional ops; nms() restates torchvision.ops.nms (torchvision 0.23.0 is the reference's pin, absent here:
PARITY UNPINNED against torchvision itself; the IoU formula is pinned by the reference's own pure-torch
box_iou, y5/utils/general.py:297-321, tests/golden/yolo_box_iou.npz)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _conv(sd, pre, x, k, s, act=True):
    """Conv.forward / fuseforward (y5/models/common.py:50-55): conv -> (BN) -> SiLU."""
    w = sd[pre + "conv.weight"]
    if pre + "conv.bias" in sd:
        y = F.conv2d(x, w, sd[pre + "conv.bias"], stride=s, padding=k // 2)
    else:
        y = F.conv2d(x, w, None, stride=s, padding=k // 2)
        y = F.batch_norm(y, sd[pre + "bn.running_mean"], sd[pre + "bn.running_var"], sd[pre + "bn.weight"],
                         sd[pre + "bn.bias"], False, 0.0, 1e-3)
    return F.silu(y) if act else y


def _bn(sd, pre, x):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"],
                        False, 0.0, 1e-3)


def _stem(sd, pre, x):
    """StemBlock.forward (common.py:67-73)."""
    s1 = _conv(sd, pre + "stem_1.", x, 3, 2)
    a = _conv(sd, pre + "stem_2a.", s1, 1, 1)
    b = _conv(sd, pre + "stem_2b.", a, 3, 2)
    p = F.max_pool2d(s1, 2, 2, ceil_mode=True)
    return _conv(sd, pre + "stem_3.", torch.cat((b, p), 1), 1, 1)


def _bottleneck(sd, pre, x, add):
    y = _conv(sd, pre + "cv2.", _conv(sd, pre + "cv1.", x, 1, 1), 3, 1)
    return x + y if add else y


def _c3(sd, pre, x, n, shortcut):
    """C3.forward (common.py:123-124)."""
    y = _conv(sd, pre + "cv1.", x, 1, 1)
    for i in range(n):
        y = _bottleneck(sd, f"{pre}m.{i}.", y, shortcut)
    return _conv(sd, pre + "cv3.", torch.cat((y, _conv(sd, pre + "cv2.", x, 1, 1)), 1), 1, 1)


def _shuffle_block(sd, pre, x, stride):
    """ShuffleV2Block.forward + channel_shuffle (common.py:21-31, 169-176)."""
    def branch2(t):
        t = F.silu(_bn(sd, pre + "branch2.1.", F.conv2d(t, sd[pre + "branch2.0.weight"])))
        t = _bn(sd, pre + "branch2.4.", F.conv2d(t, sd[pre + "branch2.3.weight"], stride=stride, padding=1,
                                                groups=t.shape[1]))
        return F.silu(_bn(sd, pre + "branch2.6.", F.conv2d(t, sd[pre + "branch2.5.weight"])))
    if stride == 1:
        x1, x2 = x.chunk(2, dim=1)
        out = torch.cat((x1, branch2(x2)), 1)
    else:
        t = _bn(sd, pre + "branch1.1.", F.conv2d(x, sd[pre + "branch1.0.weight"], stride=stride, padding=1,
                                                groups=x.shape[1]))
        t = F.silu(_bn(sd, pre + "branch1.3.", F.conv2d(t, sd[pre + "branch1.2.weight"])))
        out = torch.cat((t, branch2(x)), 1)
    b, c, h, w = out.shape
    return out.view(b, 2, c // 2, h, w).transpose(1, 2).contiguous().view(b, -1, h, w)


def _spp(sd, pre, x, ks):
    x = _conv(sd, pre + "cv1.", x, 1, 1)
    return _conv(sd, pre + "cv2.", torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in ks], 1), 1, 1)


def detect_decode(raw_heads, anchors_px, strides=(8., 16., 32.)):
    """Detect.forward inference branch (y5/models/yolo.py:68-108).  raw_heads: list of (bs, 3, ny, nx, 16)."""
    z = []
    for i, x in enumerate(raw_heads):
        bs, na, ny, nx, no = x.shape
        yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
        ag = torch.tensor(anchors_px[i]).float().view(1, na, 1, 1, 2)
        y = torch.full_like(x, 0)
        cr = [0, 1, 2, 3, 4, 15]
        y[..., cr] = x[..., cr].sigmoid()
        y[..., 5:15] = x[..., 5:15]
        y[..., 0:2] = (y[..., 0:2] * 2. - 0.5 + grid) * strides[i]
        y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * ag
        for k in range(5):
            y[..., 5 + 2 * k:7 + 2 * k] = y[..., 5 + 2 * k:7 + 2 * k] * ag + grid * strides[i]
        z.append(y.view(bs, -1, no))
    return torch.cat(z, 1)


def forward(spec, sd, x):
    """Model.forward_once (yolo.py:177-198) in eval mode -> (z (bs, n, 16), raw heads [(bs, 3, ny, nx, 16)])."""
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    gd, gw = spec["depth_multiple"], spec["width_multiple"]
    ys = []
    for i, (f, n, mname, args) in enumerate(spec["backbone"] + spec["head"]):
        n = max(round(n * gd), 1) if n > 1 else n
        if f != -1:
            x = ys[f] if isinstance(f, int) else [x if j == -1 else ys[j] for j in f]
        pre = f"model.{i}."
        if mname == "StemBlock":
            x = _stem(sd, pre, x)
        elif mname == "Conv":
            x = _conv(sd, pre, x, args[1], args[2])
        elif mname == "C3":
            x = _c3(sd, pre, x, n, args[1] if len(args) > 1 else True)
        elif mname == "ShuffleV2Block":
            for r in range(n):
                x = _shuffle_block(sd, pre if n == 1 else f"{pre}{r}.", x, args[1])
        elif mname == "SPP":
            x = _spp(sd, pre, x, args[1])
        elif mname == "nn.Upsample":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif mname == "Concat":
            x = torch.cat(x, 1)
        elif mname == "Detect":
            heads = []
            for lvl, t in enumerate(x):
                h = F.conv2d(t, sd[f"{pre}m.{lvl}.weight"], sd[f"{pre}m.{lvl}.bias"])
                bs, _, ny, nx = h.shape
                heads.append(h.view(bs, 3, 16, ny, nx).permute(0, 1, 3, 4, 2).contiguous())
            anchors_px = [np.asarray(a, dtype=np.float32).reshape(3, 2) for a in spec["anchors"]]
            return detect_decode(heads, anchors_px), heads
        ys.append(x)
    raise RuntimeError("spec has no Detect layer")


def box_iou(box1, box2):
    """y5/utils/general.py:297-321."""
    def area(b):
        return (b[2] - b[0]) * (b[3] - b[1])
    a1, a2 = area(box1.T), area(box2.T)
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms semantics: greedy over scores descending (stable), suppress IoU > threshold,
    IoU = inter / (area_i + area_j - inter) with clamped intersections; returns kept indices in score order."""
    boxes = boxes.numpy().astype(np.float32)
    order = torch.argsort(scores, descending=True, stable=True).numpy()
    x1, y1, x2, y2 = boxes.T
    areas = (x2 - x1) * (y2 - y1)
    supp = np.zeros(len(boxes), bool)
    keep = []
    for _i, i in enumerate(order):
        if supp[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1, yy1 = np.maximum(x1[i], x1[rest]), np.maximum(y1[i], y1[rest])
        xx2, yy2 = np.minimum(x2[i], x2[rest]), np.minimum(y2[i], y2[rest])
        w, h = np.maximum(np.float32(0), xx2 - xx1), np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        supp[rest[ovr > iou_threshold]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def non_max_suppression_face(prediction, conf_thres=0.25, iou_thres=0.45):
    """y5/utils/general.py:370-453 (nc = 1, classes None, agnostic False).  Returns (list of (k,16), list of kept
    row indices into the prediction rows)."""
    prediction = torch.as_tensor(prediction).clone()
    xc = prediction[..., 4] > conf_thres
    out, idxs = [], []
    for xi, x in enumerate(prediction):
        rows = torch.nonzero(xc[xi]).flatten()
        x = x[xc[xi]]
        if not x.shape[0]:
            out.append(torch.zeros((0, 16))); idxs.append(torch.zeros((0,), dtype=torch.long)); continue
        x[:, 15:] *= x[:, 4:5]
        box = x[:, :4].clone()
        box[:, 0] = x[:, 0] - x[:, 2] / 2
        box[:, 1] = x[:, 1] - x[:, 3] / 2
        box[:, 2] = x[:, 0] + x[:, 2] / 2
        box[:, 3] = x[:, 1] + x[:, 3] / 2
        conf, j = x[:, 15:].max(1, keepdim=True)
        sel = conf.view(-1) > conf_thres
        x = torch.cat((box, conf, x[:, 5:15], j.float()), 1)[sel]
        rows = rows[sel]
        if not x.shape[0]:
            out.append(torch.zeros((0, 16))); idxs.append(torch.zeros((0,), dtype=torch.long)); continue
        c = x[:, 15:16] * 4096
        i = nms(x[:, :4] + c, x[:, 4], iou_thres)
        out.append(x[i]); idxs.append(rows[i])
    return out, idxs


def w_bbox_iou(box1, box2):
    """onnx_utils.py:76-104 (x1y1x2y2, +1 pixel convention)."""
    ix1, iy1 = torch.max(box1[:, 0], box2[:, 0]), torch.max(box1[:, 1], box2[:, 1])
    ix2, iy2 = torch.min(box1[:, 2], box2[:, 2]), torch.min(box1[:, 3], box2[:, 3])
    inter = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
    a1 = (box1[:, 2] - box1[:, 0] + 1) * (box1[:, 3] - box1[:, 1] + 1)
    a2 = (box2[:, 2] - box2[:, 0] + 1) * (box2[:, 3] - box2[:, 1] + 1)
    return inter / (a1 + a2 - inter + 1e-16)


def w_non_max_suppression(prediction, conf_thres=0.5, nms_thres=0.4):
    """onnx_utils.py:107-163, num_classes = 1 (class_conf is column 5, as in the reference)."""
    prediction = torch.as_tensor(prediction).clone()
    bc = torch.zeros_like(prediction)
    bc[:, :, 0] = prediction[:, :, 0] - prediction[:, :, 2] / 2
    bc[:, :, 1] = prediction[:, :, 1] - prediction[:, :, 3] / 2
    bc[:, :, 2] = prediction[:, :, 0] + prediction[:, :, 2] / 2
    bc[:, :, 3] = prediction[:, :, 1] + prediction[:, :, 3] / 2
    prediction[:, :, :4] = bc[:, :, :4]
    output = []
    for p in prediction:
        p = p[p[:, 4] >= conf_thres]
        if not p.size(0):
            output.append(None); continue
        det = torch.cat((p[:, :5], p[:, 5:6], torch.zeros((len(p), 1))), 1)
        det = det[torch.argsort(det[:, 4], descending=True, stable=True)]
        keep = []
        while det.size(0):
            keep.append(det[0].unsqueeze(0))
            if len(det) == 1:
                break
            ious = w_bbox_iou(keep[-1], det[1:])
            det = det[1:][ious < nms_thres]
        output.append(torch.cat(keep))
    return output


def get_bboxes_confs_areas(dets, det_thres, bbox_area_thres, orig_size, in_size):
    """get_bboxes_confs_areas (fde/modules/yolov5_face/onnx/onnx_utils.py:313-340), numpy, same operation order:
    conf > det_thres; perc = 100 * area / (iw * ih) (the PERCENT, computed in the dets' dtype); perc > bbox_area_thres;
    scale_coords + round.  As in the reference the returned perc is NOT filtered by the area test (it has one entry
    per detection that passed the confidence test).  Pinned by tests/golden/yolo_bboxes_confs_areas.npz."""
    from .image_ref import scale_coords
    w, h = orig_size
    iw, ih = in_size
    dets = np.asarray(dets)
    dets = dets[dets[..., 4] > det_thres]
    total_area = iw * ih
    bbox_area = (dets[:, 2] - dets[:, 0]) * (dets[:, 3] - dets[:, 1])
    bbox_area_perc = 100 * bbox_area / total_area
    dets = dets[bbox_area_perc > bbox_area_thres]
    boxes = scale_coords((ih, iw), dets[..., :4].copy(), (h, w)).round()
    return boxes, dets[..., 4], bbox_area_perc

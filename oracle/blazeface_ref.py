"""Oracle for BlazeFace (fde/modules/blazeface/blazeface.py).  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch
import torch.nn.functional as F


def _blaze_block(sd, pre, x, stride):
    """BlazeBlock.forward (blazeface.py:36-47)."""
    wd, bd = sd[pre + "convs.0.weight"], sd[pre + "convs.0.bias"]
    wp, bp = sd[pre + "convs.1.weight"], sd[pre + "convs.1.bias"]
    cin, cout = wd.shape[0], wp.shape[0]
    if stride == 2:
        h = F.pad(x, (0, 2, 0, 2))
        sc = F.max_pool2d(x, 2, 2)
        pad = 0
    else:
        h, sc, pad = x, x, 1
    if cout > cin:
        sc = F.pad(sc, (0, 0, 0, 0, 0, cout - cin))
    y = F.conv2d(F.conv2d(h, wd, bd, stride=stride, padding=pad, groups=cin), wp, bp)
    return F.relu(y + sc)


BACK_SPEC = ([1] * 7 + [2] + [1] * 7 + [2] + [1] * 7 + [2] + [1] * 7)          # blazeface.py:122-152
FRONT_SPEC1 = [1, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1]                                  # blazeface.py:166-176
FRONT_SPEC2 = [2, 1, 1, 1, 1]                                                    # blazeface.py:180-184


def forward(sd, x, back_model):
    """BlazeFace.forward (blazeface.py:192-228): x (b,3,H,W) float in [-1,1] -> r (b,896,16), c (b,896,1)."""
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    b = x.shape[0]
    x = F.pad(x, (1, 2, 1, 2))
    name = "backbone" if back_model else "backbone1"
    x = F.relu(F.conv2d(x, sd[f"{name}.0.weight"], sd[f"{name}.0.bias"], stride=2))
    for i, s in enumerate(BACK_SPEC if back_model else FRONT_SPEC1):
        x = _blaze_block(sd, f"{name}.{i + 2}.", x, s)
    if back_model:
        h = F.pad(x, (0, 2, 0, 2))                                                # FinalBlazeBlock :65-68
        h = F.conv2d(h, sd["final.convs.0.weight"], sd["final.convs.0.bias"], stride=2, groups=h.shape[1])
        h = F.relu(F.conv2d(h, sd["final.convs.1.weight"], sd["final.convs.1.bias"]))
    else:
        h = x
        for i, s in enumerate(FRONT_SPEC2):
            h = _blaze_block(sd, f"backbone2.{i}.", h, s)

    def head(t, nm, last):
        y = F.conv2d(t, sd[nm + ".weight"], sd[nm + ".bias"])
        return y.permute(0, 2, 3, 1).reshape(b, -1, last)

    c = torch.cat((head(x, "classifier_8", 1), head(h, "classifier_16", 1)), dim=1)
    r = torch.cat((head(x, "regressor_8", 16), head(h, "regressor_16", 16)), dim=1)
    return r, c


def decode_boxes(raw, anchors, scale):
    """_decode_boxes (blazeface.py:373-402); scale = x/y/w/h_scale (all equal, :99-109)."""
    raw, anchors = torch.as_tensor(raw), torch.as_tensor(anchors)
    boxes = torch.zeros_like(raw)
    xc = raw[..., 0] / scale * anchors[:, 2] + anchors[:, 0]
    yc = raw[..., 1] / scale * anchors[:, 3] + anchors[:, 1]
    w = raw[..., 2] / scale * anchors[:, 2]
    h = raw[..., 3] / scale * anchors[:, 3]
    boxes[..., 0] = yc - h / 2.
    boxes[..., 1] = xc - w / 2.
    boxes[..., 2] = yc + h / 2.
    boxes[..., 3] = xc + w / 2.
    for k in range(6):
        o = 4 + 2 * k
        boxes[..., o] = raw[..., o] / scale * anchors[:, 2] + anchors[:, 0]
        boxes[..., o + 1] = raw[..., o + 1] / scale * anchors[:, 3] + anchors[:, 1]
    return boxes


def tensors_to_detections(raw_box, raw_score, anchors, scale, clip=100.0, min_score=0.65):
    """_tensors_to_detections (blazeface.py:321-371): list of (n_i, 17) per image, anchor order kept."""
    raw_box, raw_score = torch.as_tensor(raw_box), torch.as_tensor(raw_score)
    boxes = decode_boxes(raw_box, anchors, scale)
    scores = raw_score.clamp(-clip, clip).sigmoid().squeeze(-1)
    mask = scores >= min_score
    return [torch.cat((boxes[i, mask[i]], scores[i, mask[i]].unsqueeze(-1)), dim=-1) for i in range(raw_box.shape[0])]


def overlap_similarity(box, others):
    """intersect/jaccard/overlap_similarity (blazeface.py:463-526) for one box against (n,4)."""
    max_xy = torch.min(box[2:].unsqueeze(0), others[:, 2:])
    min_xy = torch.max(box[:2].unsqueeze(0), others[:, :2])
    inter = torch.clamp(max_xy - min_xy, min=0)
    inter = inter[:, 0] * inter[:, 1]
    area_a = (box[2] - box[0]) * (box[3] - box[1])
    area_b = (others[:, 2] - others[:, 0]) * (others[:, 3] - others[:, 1])
    return inter / (area_a + area_b - inter)


def weighted_nms(dets, thr=0.3):
    """_weighted_non_max_suppression (blazeface.py:404-458).  Returns (out (k,17), member_of (n,)).
    Sort is descending by score, ties by index (torch.argsort(..., stable=True)); the reference's
    argsort is unstable, so fixtures avoid exact ties.  A remaining[0] whose self-IoU is not > thr
    (degenerate box; the reference loops forever) is emitted alone and removed."""
    dets = torch.as_tensor(dets, dtype=torch.float32)
    n = dets.shape[0]
    member = torch.full((n,), -1, dtype=torch.int32)
    if n == 0:
        return torch.zeros((0, 17)), member
    remaining = torch.argsort(dets[:, 16], descending=True, stable=True)
    out = []
    while len(remaining) > 0:
        det = dets[remaining[0]]
        ious = overlap_similarity(det[:4], dets[remaining, :4])
        mask = ious > thr
        mask[0] = True
        overlapping = remaining[mask]
        remaining = remaining[~mask]
        w = det.clone()
        if len(overlapping) > 1:
            coords = dets[overlapping, :16]
            scores = dets[overlapping, 16:17]
            total = scores.sum()
            w[:16] = (coords * scores).sum(0) / total
            w[16] = total / len(overlapping)
        member[overlapping] = len(out)
        out.append(w)
    return torch.stack(out), member


def predict_on_batch(sd, x_u8_nchw, anchors, back_model):
    """predict_on_batch (blazeface.py:266-319) on a u8/float NCHW RGB batch already at model size."""
    scale = 256.0 if back_model else 128.0
    x = torch.as_tensor(x_u8_nchw).float() / 127.5 - 1.0                          # _preprocess :248-250
    with torch.no_grad():
        r, c = forward(sd, x, back_model)
    dets = tensors_to_detections(r, c, anchors, scale, 100.0, 0.65 if back_model else 0.75)
    return [weighted_nms(d, 0.3)[0] for d in dets], (r, c)

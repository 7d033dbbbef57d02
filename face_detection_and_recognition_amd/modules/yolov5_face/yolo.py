"""YOLOv5-face on MI355X: the reference's module classes, yaml-driven model builder and ``state_dict`` keys
(face_detection_and_extraction/modules/yolov5_face/pytorch/models/{common,yolo,experimental}.py) compiled
to a HIP plan.

  Conv / fuse_conv_and_bn   common.py:39-55, utils/torch_utils.py:164-184  -> FP_OP_CONV (+BN affine or folded bias, SiLU)
  StemBlock                 common.py:58-73
  Bottleneck / C3           common.py:76-87, 111-124
  ShuffleV2Block            common.py:21-31, 127-176 (BatchNorm stays live: Model.fuse only folds `Conv`)
  SPP / Upsample / Concat   common.py:179-191, 235-242
  Detect                    yolo.py:29-113  -> FP_OP_CONV heads + fp_yolo_decode
  Model / parse_model       yolo.py:116-327 (the reference's parse_model is broken for its own yamls, SURVEY F5;
                            this one resolves module names and nc/anchors from the spec directly)
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from ... import _lib as L
from ...plan import CompiledPlan, PlanBuilder, PlanCache, View, bn_affine, cpad, switch_key
from ..params import BNParams, ConvParams, _NoCompute, npy

# Architecture specs of the reference's in-tree yamls (y5/models/yolov5n.yaml, yolov5s.yaml, yolov5n-0.5.yaml):
# [from, number, module, args] rows, same meaning as upstream.
ANCHORS = [[4, 5, 8, 10, 13, 16], [23, 29, 43, 55, 73, 105], [146, 217, 231, 300, 335, 433]]
_HEAD_N = [[-1, 1, "Conv", [128, 1, 1]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 4], 1, "Concat", [1]],
           [-1, 1, "C3", [128, False]], [-1, 1, "Conv", [128, 1, 1]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]],
           [[-1, 2], 1, "Concat", [1]], [-1, 1, "C3", [128, False]], [-1, 1, "Conv", [128, 3, 2]],
           [[-1, 11], 1, "Concat", [1]], [-1, 1, "C3", [128, False]], [-1, 1, "Conv", [128, 3, 2]],
           [[-1, 7], 1, "Concat", [1]], [-1, 1, "C3", [128, False]], [[14, 17, 20], 1, "Detect", ["nc", "anchors"]]]
_BACKBONE_N = [[-1, 1, "StemBlock", [32, 3, 2]], [-1, 1, "ShuffleV2Block", [128, 2]], [-1, 3, "ShuffleV2Block", [128, 1]],
               [-1, 1, "ShuffleV2Block", [256, 2]], [-1, 7, "ShuffleV2Block", [256, 1]],
               [-1, 1, "ShuffleV2Block", [512, 2]], [-1, 3, "ShuffleV2Block", [512, 1]]]
SPECS = {
    "yolov5n": dict(nc=1, depth_multiple=1.0, width_multiple=1.0, anchors=ANCHORS, backbone=_BACKBONE_N, head=_HEAD_N),
    "yolov5n-0.5": dict(nc=1, depth_multiple=1.0, width_multiple=0.5, anchors=ANCHORS, backbone=_BACKBONE_N,
                        head=_HEAD_N),
    "yolov5s": dict(
        nc=1, depth_multiple=0.33, width_multiple=0.35, anchors=ANCHORS,
        backbone=[[-1, 1, "StemBlock", [64, 3, 2]], [-1, 3, "C3", [128]], [-1, 1, "Conv", [256, 3, 2]],
                  [-1, 9, "C3", [256]], [-1, 1, "Conv", [512, 3, 2]], [-1, 9, "C3", [512]],
                  [-1, 1, "Conv", [1024, 3, 2]], [-1, 1, "SPP", [1024, [3, 5, 7]]], [-1, 3, "C3", [1024, False]]],
        head=[[-1, 1, "Conv", [512, 1, 1]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 5], 1, "Concat", [1]],
              [-1, 3, "C3", [512, False]], [-1, 1, "Conv", [256, 1, 1]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]],
              [[-1, 3], 1, "Concat", [1]], [-1, 3, "C3", [256, False]], [-1, 1, "Conv", [256, 3, 2]],
              [[-1, 13], 1, "Concat", [1]], [-1, 3, "C3", [512, False]], [-1, 1, "Conv", [512, 3, 2]],
              [[-1, 9], 1, "Concat", [1]], [-1, 3, "C3", [1024, False]], [[16, 19, 22], 1, "Detect", ["nc", "anchors"]]]),
}


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def autopad(k, p=None):
    """common.py:14-18."""
    return k // 2 if p is None else p


def fuse_conv_and_bn_arrays(w, bn):
    """fuse_conv_and_bn (utils/torch_utils.py:164-184) in torch fp32, same operation order."""
    w_conv = w.clone().view(w.shape[0], -1)
    w_bn = torch.diag(bn.weight.div(torch.sqrt(bn.eps + bn.running_var)))
    fw = torch.mm(w_bn, w_conv).view(w.shape)
    b_conv = torch.zeros(w.shape[0])
    b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
    fb = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn
    return fw, fb


def _bn_sb(bn):
    return bn_affine(npy(bn.weight), npy(bn.bias), npy(bn.running_mean), npy(bn.running_var), bn.eps)


class Conv(_NoCompute):
    """conv(no bias) -> BN -> SiLU; after fuse(): conv(with bias) -> SiLU (common.py:39-55)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        assert g == 1
        self.c1, self.c2, self.k, self.s, self.p = c1, c2, k, s, autopad(k, p)
        self.conv = ConvParams(c1, c2, k, s, self.p, bias=False)
        self.bn = BNParams(c2, eps=1e-3)      # initialize_weights sets eps=1e-3 (utils/torch_utils.py:102-104)
        self.act = act

    def fuse(self):
        if self.bn is None:
            return
        fw, fb = fuse_conv_and_bn_arrays(self.conv.weight.detach().cpu(), self.bn.cpu())
        dev = self.conv.weight.device
        self.conv.weight = nn.Parameter(fw.to(dev), requires_grad=False)
        self.conv.bias = nn.Parameter(fb.to(dev), requires_grad=False)
        self.bn = None

    def make_fused_structure(self):
        if self.bn is not None:
            self.conv.bias = nn.Parameter(torch.zeros(self.c2, device=self.conv.weight.device), requires_grad=False)
            self.bn = None

    def emit(self, pb, x, out=None, res=None):
        OH = (x.H + 2 * self.p - self.k) // self.s + 1
        OW = (x.W + 2 * self.p - self.k) // self.s + 1
        if out is None:
            out = pb.new_buf(OH, OW, self.c2).view()
        kw = dict(stride=self.s, pad=(self.p, self.p), act=L.ACT_SILU if self.act else L.ACT_NONE)
        if self.bn is None:
            kw["bias"] = npy(self.conv.bias)
        else:
            kw["scale"], kw["bias"] = _bn_sb(self.bn)
        if res is not None:
            kw.update(res=res, res_mode=L.RES_ADD_AFTER_ACT)
        pb.conv(x, npy(self.conv.weight), out, **kw)
        return out


class StemBlock(_NoCompute):
    """common.py:58-73."""

    def __init__(self, c1, c2, k=3, s=2, p=None, g=1, act=True):
        super().__init__()
        self.c2 = c2
        self.stem_1 = Conv(c1, c2, k, s, p, g, act)
        self.stem_2a = Conv(c2, c2 // 2, 1, 1, 0)
        self.stem_2b = Conv(c2 // 2, c2, 3, 2, 1)
        self.stem_3 = Conv(c2 * 2, c2, 1, 1, 0)

    FUSE = True   # class-wide switch: False emits the five separate ops (A/B parity tests)
    FUSE_TAIL = True   # stem_2b + cat + stem_3 as one FP_OP_YSTEM2 (csrc/ystem2.hip) where the block has its shape (c = 32)

    @staticmethod
    def _sb(conv):
        """(scale, bias) of a Conv's epilogue: live BatchNorm affine, or (None, folded bias) after fuse()."""
        if conv.bn is None:
            return None, npy(conv.conv.bias)
        return _bn_sb(conv.bn)

    def fusable(self, H, W):
        """FP_OP_YSTEM handles the block's head (stem_1 -> LDS -> stem_2a + maxpool) for these shapes."""
        c, s1c = self.c2, self.stem_1
        return (StemBlock.FUSE and H % 4 == 0 and W % 4 == 0 and c <= 32 and c % 8 == 0 and s1c.k == 3 and s1c.s == 2 and
                s1c.p == 1 and s1c.act and self.stem_2a.act)

    def emit(self, pb, x, out=None, u8=None):
        """u8 = (H, W, frame_h, frame_w, ext_index): the block reads the u8 frames itself (letterbox fused into the
        staging, FP_OP_YSTEM_U8) and x is None."""
        c = self.c2
        s1c = self.stem_1
        if u8 is not None or (x.C == 4 and x.buf.ld == 4 and x.coff == 0 and self.fusable(x.H, x.W)):
            # stem_1 -> (LDS) -> stem_2a + maxpool in one kernel (FP_OP_YSTEM): stem_1's output never reaches HBM
            H1, W1 = (x.H // 2, x.W // 2) if u8 is None else (u8[0] // 2, u8[1] // 2)
            a = pb.new_buf(H1, W1, c // 2)
            cat = pb.new_buf(H1 // 2, W1 // 2, 2 * c)
            sc1, bi1 = self._sb(s1c)
            sc2, bi2 = self._sb(self.stem_2a)
            pb.ystem(x, npy(s1c.conv.weight), sc1, bi1, npy(self.stem_2a.conv.weight), sc2, bi2, a.view(0, cpad(c // 2)),
                     cat.view(c, c), u8=u8)
            s2b, s3 = self.stem_2b, self.stem_3
            y = pb.new_buf(H1 // 2, W1 // 2, c).view() if out is None else out
            if (StemBlock.FUSE_TAIL and c == 32 and s2b.act and s3.act and s2b.k == 3 and s2b.s == 2 and s2b.p == 1 and
                    pb.ystem2_supported(a.view(0, c // 2), cat.view(c, c), y)):
                # stem_2b -> cat -> stem_3 in one kernel (FP_OP_YSTEM2): stem_2b's output never reaches HBM
                pb.ystem2(a.view(0, c // 2), cat.view(c, c), npy(s2b.conv.weight), self._sb(s2b), npy(s3.conv.weight), self._sb(s3), y)
                pb.free(a)
                pb.free(cat)
                return y
            if out is None:
                pb.free(y.buf)
            self.stem_2b.emit(pb, a.view(0, c // 2), out=cat.view(0, c))
            pb.free(a)
            y = self.stem_3.emit(pb, cat.view(), out=out)
            pb.free(cat)
            return y
        s1 = self.stem_1.emit(pb, x)
        a = self.stem_2a.emit(pb, s1)
        OH, OW = math.ceil(s1.H / 2), math.ceil(s1.W / 2)            # MaxPool2d(2, 2, ceil_mode=True)
        cat = pb.new_buf(OH, OW, 2 * self.c2)
        self.stem_2b.emit(pb, a, out=cat.view(0, self.c2))
        pb.maxpool(s1, cat.view(self.c2, self.c2), 2, 2, 0)
        pb.free(a.buf)
        pb.free(s1.buf)
        y = self.stem_3.emit(pb, cat.view(), out=out)
        pb.free(cat)
        return y


class Bottleneck(_NoCompute):
    """common.py:76-87."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def emit(self, pb, x, out=None):
        t = self.cv1.emit(pb, x)
        y = self.cv2.emit(pb, t, out=out, res=x if self.add else None)
        pb.free(t.buf)
        return y


class C3(_NoCompute):
    """common.py:111-124: cv3(cat(m(cv1(x)), cv2(x)))."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.c_, self.c2 = c_, c2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    MERGE = True   # class-wide switch: False emits cv1 and cv2 as separate convs (A/B parity tests)

    def emit(self, pb, x, out=None):
        assert self.c_ % 4 == 0, "C3 hidden width must be a multiple of 4 for the in-place concat"
        c_ = self.c_
        cat = pb.new_buf(x.H, x.W, 2 * c_)
        a, b = self.cv1, self.cv2
        if C3.MERGE and (a.bn is None) == (b.bn is None) and a.act and b.act:
            # cv1 and cv2 are two 1x1 convs on the same input whose outputs end up side by side in the concat: ONE conv
            # with the weights stacked writes whole concat rows (x is read once; a conv that writes only one half of
            # every row ran at half the store rate: 489 vs 232 us for 24 -> 24 at 160x160, profiles/r02_yolo).  The
            # bottleneck chain then updates the first half in place (residual and output are the same view).
            w = np.concatenate([npy(a.conv.weight), npy(b.conv.weight)], 0)
            if a.bn is None:
                kw = dict(bias=np.concatenate([npy(a.conv.bias), npy(b.conv.bias)]))
            else:
                (s1, b1), (s2, b2) = _bn_sb(a.bn), _bn_sb(b.bn)
                kw = dict(scale=np.concatenate([s1, s2]), bias=np.concatenate([b1, b2]))
            pb.conv(x, w, cat.view(0, 2 * c_), act=L.ACT_SILU, n_convs=2, **kw)
            t = cat.view(0, c_)
            for blk in self.m:
                blk.emit(pb, t, out=t)
        else:
            t = a.emit(pb, x)
            for i, blk in enumerate(self.m):
                last = i == len(self.m) - 1
                y = blk.emit(pb, t, out=cat.view(0, c_) if last else None)
                pb.free(t.buf)
                t = y
            b.emit(pb, x, out=cat.view(c_, c_))
        out = self.cv3.emit(pb, cat.view(), out=out)
        pb.free(cat)
        return out


class _Tag(_NoCompute):
    """Parameter-free placeholder (nn.SiLU slots in ShuffleV2Block branches) so child indices match."""


class ShuffleV2Block(_NoCompute):
    """common.py:127-176 + channel_shuffle (:21-31).  The concat + shuffle is pure addressing: the two halves are
    written interleaved (out_cmul = 2) straight into the output tensor."""

    def __init__(self, inp, oup, stride):
        super().__init__()
        self.stride, self.inp, self.oup = stride, inp, oup
        bf = oup // 2
        self.bf = bf
        assert stride != 1 or inp == bf << 1
        if stride > 1:
            self.branch1 = nn.Sequential(ConvParams(inp, inp, 3, stride, 1, groups=inp, bias=False), BNParams(inp, eps=1e-3),
                                         ConvParams(inp, bf, 1, 1, 0, bias=False), BNParams(bf, eps=1e-3), _Tag())
        else:
            self.branch1 = nn.Sequential()
        c_in2 = inp if stride > 1 else bf
        self.branch2 = nn.Sequential(ConvParams(c_in2, bf, 1, 1, 0, bias=False), BNParams(bf, eps=1e-3), _Tag(),
                                     ConvParams(bf, bf, 3, stride, 1, groups=bf, bias=False), BNParams(bf, eps=1e-3),
                                     ConvParams(bf, bf, 1, 1, 0, bias=False), BNParams(bf, eps=1e-3), _Tag())

    FUSE = True   # class-wide switch: False emits the unfused DWCONV + CONV pairs (A/B parity tests)
    FUSE_DOWN = True   # the whole stride-2 block as one FP_OP_SHUFDOWN where csrc/shufdown.hip has the shape (32 -> 128 channels)
    FUSE_UNIT = True   # the whole stride-1 block as one FP_OP_SHUFUNIT where it has the width (128 channels)

    def emit(self, pb, x, out=None):
        """out: optional View (C = oup) the block writes into, e.g. a channel slice of a later Concat's buffer."""
        s = self.stride
        OH, OW = (x.H + 2 - 3) // s + 1, (x.W + 2 - 3) // s + 1
        out = pb.new_buf(OH, OW, self.oup).view() if out is None else out
        assert out.C == self.oup and out.cmul == 1
        ob, oc = out.buf, out.coff
        # cat + channel_shuffle(2) is the last conv's epilogue (FP_RES_SHUFFLE2): out[2n] = other half, out[2n+1] = conv
        b2 = self.branch2
        fuse2 = ShuffleV2Block.FUSE and self.bf % 64 == 0 and self.bf <= 128     # dw3x3 + 1x1 of branch2 in one kernel
        if s == 2 and ShuffleV2Block.FUSE and ShuffleV2Block.FUSE_DOWN and pb.shufdown_supported(x, View(ob, oc, self.oup), self.inp, self.bf):
            b1 = self.branch1
            pb.shufdown(x, npy(b1[0].weight), _bn_sb(b1[1]), npy(b1[2].weight), _bn_sb(b1[3]),
                        npy(b2[0].weight), _bn_sb(b2[1]), npy(b2[3].weight), _bn_sb(b2[4]), npy(b2[5].weight), _bn_sb(b2[6]),
                        View(ob, oc, self.oup))
            return out
        if s == 1 and ShuffleV2Block.FUSE and ShuffleV2Block.FUSE_UNIT and pb.shufunit_supported(x, View(ob, oc, self.oup), self.bf):
            pb.shufunit(x, npy(b2[0].weight), _bn_sb(b2[1]), npy(b2[3].weight), _bn_sb(b2[4]), npy(b2[5].weight), _bn_sb(b2[6]),
                        View(ob, oc, self.oup))
            return out
        if s == 1:
            first = View(x.buf, x.coff, self.bf)                                          # x1 passthrough
            x2 = View(x.buf, x.coff + self.bf, self.bf)
            b1out = None
        else:
            b1 = self.branch1
            b1out = pb.new_buf(OH, OW, self.bf)
            sc, bi = _bn_sb(b1[1])
            sc2, bi2 = _bn_sb(b1[3])
            if ShuffleV2Block.FUSE and self.inp % 64 == 0 and self.bf <= 128:
                # branch1: dw3x3 s2 + BN -> 1x1 + BN + SiLU as one FP_OP_DWPW (the depthwise result stays in LDS)
                pb.dwpw(x, npy(b1[0].weight), sc, bi, None, npy(b1[2].weight), sc2, bi2, b1out.view(), s,
                        out_act=L.ACT_SILU)
            else:
                t = pb.new_buf(OH, OW, self.inp)
                pb.dwconv(x, npy(b1[0].weight), t.view(), stride=s, pad=(1, 1), scale=sc, bias=bi)
                pb.conv(t.view(), npy(b1[2].weight), b1out.view(), scale=sc2, bias=bi2, act=L.ACT_SILU)
                pb.free(t)
            first = b1out.view()
            x2 = x
        t1 = pb.new_buf(x.H, x.W, self.bf)
        sc, bi = _bn_sb(b2[1])
        pb.conv(x2, npy(b2[0].weight), t1.view(), scale=sc, bias=bi, act=L.ACT_SILU)
        sc, bi = _bn_sb(b2[4])
        sc2, bi2 = _bn_sb(b2[6])
        if fuse2:
            # branch2 tail: dw3x3 + BN -> 1x1 + BN + SiLU -> cat + channel_shuffle in one kernel (FP_OP_DWPW with the
            # FP_RES_SHUFFLE2 epilogue): the depthwise tensor never reaches HBM
            pb.dwpw(t1.view(), npy(b2[3].weight), sc, bi, None, npy(b2[5].weight), sc2, bi2, View(ob, oc, self.bf), s,
                    res=first, out_act=L.ACT_SILU, shuffle=True)
            pb.free(t1)
        else:
            t2 = pb.new_buf(OH, OW, self.bf)
            pb.dwconv(t1.view(), npy(b2[3].weight), t2.view(), stride=s, pad=(1, 1), scale=sc, bias=bi)
            pb.free(t1)
            if self.bf % 4 == 0:
                pb.conv(t2.view(), npy(b2[5].weight), View(ob, oc, self.bf), scale=sc2, bias=bi2, act=L.ACT_SILU,
                        res=first, res_mode=L.RES_SHUFFLE2)
            else:   # odd widths: scalar interleaved writes
                pb.copy(first, View(ob, oc, self.bf, cmul=2))
                pb.conv(t2.view(), npy(b2[5].weight), View(ob, oc + 1, self.bf, cmul=2), scale=sc2, bias=bi2,
                        act=L.ACT_SILU)
            pb.free(t2)
        if b1out is not None:
            pb.free(b1out)
        return out


class SPP(_NoCompute):
    """common.py:179-191."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        c_ = c1 // 2
        self.c_, self.k = c_, tuple(k)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)

    def emit(self, pb, x, out=None):
        assert self.c_ % 4 == 0
        cat = pb.new_buf(x.H, x.W, self.c_ * (len(self.k) + 1))
        first = cat.view(0, self.c_)
        self.cv1.emit(pb, x, out=first)
        prev, prev_k = first, 1
        for i, k in enumerate(self.k):
            dst = cat.view(self.c_ * (i + 1), self.c_)
            if k - prev_k == 2:     # stride-1 max pools cascade exactly: pool_k = pool_3(pool_{k-2}) (max is associative)
                pb.maxpool(prev, dst, 3, 1, 1)
            else:
                pb.maxpool(first, dst, k, 1, k // 2)
            prev, prev_k = dst, k
        y = self.cv2.emit(pb, cat.view(), out=out)
        pb.free(cat)
        return y


class Upsample(_NoCompute):
    def emit(self, pb, x, out=None):
        out = pb.new_buf(2 * x.H, 2 * x.W, x.C).view(0, x.C) if out is None else out
        return pb.upsample2x(x, out)


class Concat(_NoCompute):
    """common.py:235-242 (dimension 1)."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def emit(self, pb, xs, out=None):
        """out: the buffer view the producers were told to write into (Model._emit places concat inputs at their
        channel offset when they are produced, so the concat is pure addressing); inputs that are not already in
        place are copied."""
        C = sum(v.C for v in xs)
        out = pb.new_buf(xs[0].H, xs[0].W, C).view() if out is None else out
        off = 0
        for v in xs:
            assert v.C % 4 == 0
            if not (v.buf is out.buf and v.coff == out.coff + off and v.cmul == 1):
                pb.copy(v, View(out.buf, out.coff + off, v.C))
            off += v.C
        return out


class Detect(_NoCompute):
    """yolo.py:29-113: per level a 1x1 conv to na*(nc+5+10) channels, then the inference decode."""
    stride = None

    def __init__(self, nc=1, anchors=(), ch=()):
        super().__init__()
        self.nc = nc
        self.no = nc + 5 + 10
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer("anchors", a)
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(ConvParams(x, self.no * self.na, 1, bias=True) for x in ch)


class Model(nn.Module):
    """yolo.py:116-257.  ``forward(x)``: (b, 3, H, W) float in [0, 1], H and W multiples of 32 ->
    ``(z, heads)`` with z (b, sum(3*ny*nx), 16) decoded predictions, like the reference in eval mode."""

    def __init__(self, cfg="yolov5s", ch=3, nc=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = cfg
        else:
            name = os.path.basename(str(cfg))
            for ext in (".yaml", ".yml"):
                if name.endswith(ext):
                    name = name[:-len(ext)]
            if name not in SPECS:
                raise FileNotFoundError(f"unknown YOLOv5-face config {cfg!r}; known: {sorted(SPECS)}")
            self.yaml = SPECS[name]
        if nc and nc != self.yaml["nc"]:
            self.yaml = dict(self.yaml, nc=nc)
        self.model, self.save, self.ch_out = parse_model(self.yaml, [ch])
        m = self.model[-1]
        # strides of the three levels (the reference measures them with a dry forward pass, yolo.py:140-146)
        m.stride = torch.tensor([8., 16., 32.])
        m.anchors = m.anchors / m.stride.view(-1, 1, 1)
        self.stride = m.stride
        self.names = [str(i) for i in range(self.yaml["nc"])]
        self._plans = PlanCache()

    # ---- reference API ----
    def fuse(self):
        for mod in self.model.modules():
            if isinstance(mod, Conv):
                mod.fuse()
        self._plans.clear()
        return self

    def float(self):
        return self

    def load_state_dict(self, sd, *a, **k):
        if any(key.endswith(".conv.bias") for key in sd) and not any(".stem_1.bn." in key for key in sd):
            for mod in self.model.modules():            # checkpoint of a fused model (Conv without bn)
                if isinstance(mod, Conv):
                    mod.make_fused_structure()
        out = super().load_state_dict(sd, *a, **k)
        self._plans.clear()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._plans.clear()
        return out

    def _device(self):
        return self.model[-1].m[0].weight.device

    # ---- plan ----
    CONCAT_IN_PLACE = True   # class-wide switch: False copies every Concat input (A/B parity tests)
    FOLD_UPSAMPLE = True     # class-wide switch: False launches upsample2x_kernel in front of every Concat (A/B parity tests)

    FUSE_LETTERBOX = True    # class-wide switch: False keeps the stand-alone letterbox kernel (A/B parity tests)

    def letterbox_fusable(self, H, W):
        first = self.model[0]
        return Model.FUSE_LETTERBOX and isinstance(first, StemBlock) and first.fusable(H, W)

    def _emit(self, N, H=640, W=640, frame_hw=None):
        """frame_hw = (frame_h, frame_w): the plan reads u8 frames of that size directly (no fp32 canvas, no letterbox
        kernel: external buffers 0..2 = frames, tap tables, LUT); None: the plan input is the NHWC fp32 canvas."""
        assert H % 32 == 0 and W % 32 == 0
        pb = PlanBuilder(N)
        inp = pb.new_buf(H, W, 3) if frame_hw is None else None
        det = self.model[-1]
        layers = list(self.model)

        def srcs(m):
            return [m.i - 1 if j == -1 else j for j in ([m.f] if isinstance(m.f, int) else m.f)]

        # liveness: layer j's output dies after the last layer that reads it
        last_use = {}
        for m in layers:
            for src in srcs(m):
                if src >= 0:
                    last_use[src] = m.i
        # Concat as pure addressing (common.py:235-242): every input of a Concat is produced straight into its channel
        # slice of the concat buffer, which is allocated when the first of them is produced (at most one Concat per
        # producer; an input that feeds two Concats is copied into the second)
        place = {}        # producer layer index -> (concat layer index, position among its inputs)
        if Model.CONCAT_IN_PLACE:
            for m in layers:
                if isinstance(m, Concat):
                    for pos, src in enumerate(srcs(m)):
                        prod = layers[src]
                        last = prod[-1] if isinstance(prod, nn.Sequential) else prod
                        if src >= 0 and src not in place and isinstance(last, (Conv, C3, ShuffleV2Block, SPP, StemBlock,
                                                                              Upsample)):
                            place[src] = (m.i, pos)
        cat_bufs = {}     # concat layer index -> Buf
        # nn.Upsample in front of a Concat (first input) whose readers are pointwise convs on the split-MFMA kernel: the
        # upsampled slice is never written, the readers take those channels from the half-size map (FP_OPF_IN_UP2).
        # fold[concat layer] = the half-size view; its producer then lives as long as the Concat's output does.
        fold = {}
        fold_src = {}     # Upsample layer index -> concat layer index
        if Model.CONCAT_IN_PLACE and Model.FOLD_UPSAMPLE and PlanBuilder.X6 and PlanBuilder.UP2_FOLD:
            for m in layers:
                if isinstance(m, Upsample) and place.get(m.i, (None, None))[1] == 0 and srcs(m)[0] >= 0:
                    cat_i = place[m.i][0]
                    fold_src[m.i] = cat_i
                    s0 = srcs(m)[0]
                    last_use[s0] = max(last_use.get(s0, s0), last_use.get(cat_i, cat_i))

        outs = {}
        heads = []
        for m in layers:
            fs = [m.f] if isinstance(m.f, int) else m.f
            ins = [(inp.view() if inp is not None else None) if (j == -1 and m.i == 0) else outs[m.i - 1 if j == -1 else j]
                   for j in fs]
            if isinstance(m, Detect):
                for src, conv in zip(ins, m.m):
                    hb = pb.new_buf(src.H, src.W, det.no * det.na)
                    pb.conv(src, npy(conv.weight), hb.view(0, det.no * det.na), bias=npy(conv.bias))
                    heads.append(hb)
                break
            target = None
            if isinstance(m, Concat):
                if m.i in cat_bufs:
                    target = cat_bufs[m.i].view()
            elif m.i in place:
                cat_i, pos = place[m.i]
                c_self = self.ch_out[m.i]
                if cat_i not in cat_bufs:
                    widths = [self.ch_out[src] for src in srcs(layers[cat_i])]     # parse_model's channel table
                    if any(w % 4 for w in widths):
                        del place[m.i]              # unaligned width: fall back to a copy
                    else:
                        oh, ow = self._out_hw(m, ins[0] if ins[0] is not None else pb.new_shape(H, W))
                        cat_bufs[cat_i] = pb.new_buf(oh, ow, sum(widths))
                        cat_bufs[cat_i].widths = widths
                if m.i in place:
                    cb = cat_bufs[cat_i]
                    target = View(cb, sum(cb.widths[:pos]), c_self)
            xin = ins if isinstance(m, Concat) else ins[0]
            if m.i in fold_src and target is not None and m.i in place:
                # folded Upsample: nothing is launched; the Concat's output will carry the half-size view
                fold[fold_src[m.i]] = xin
                y = target
            elif isinstance(m, nn.Sequential):
                y = xin
                for k, sub in enumerate(m):
                    y2 = sub.emit(pb, y, out=target) if (target is not None and k == len(m) - 1) else sub.emit(pb, y)
                    if y is not xin:
                        pb.free(y.buf)
                    y = y2
            elif m.i == 0 and frame_hw is not None:
                y = m.emit(pb, None, out=target, u8=(H, W, frame_hw[0], frame_hw[1], 0))
            else:
                y = m.emit(pb, xin, out=target) if target is not None else m.emit(pb, xin)
            if isinstance(m, Concat) and m.i in fold:
                assert y.buf is cat_bufs[m.i] and y.coff == 0
                y.up = fold[m.i]
            outs[m.i] = y
            for j, lu in last_use.items():
                if lu == m.i and j in outs and not any(outs[j].buf is cb for cb in cat_bufs.values()):
                    pb.free(outs[j].buf)
            for cat_i, cb in list(cat_bufs.items()):
                # a concat buffer dies after the last reader of the Concat's output AND of every input placed in it
                members = [cat_i] + [src for src, (ci, _) in place.items() if ci == cat_i]
                if m.i == max(last_use.get(j, j) for j in members) and not getattr(cb, "freed", False):
                    pb.free(cb)
                    cb.freed = True
        n_rows = sum(det.na * hb.H * hb.W for hb in heads)
        z_off, _ = pb.new_raw(n_rows * det.no)
        return pb, inp, heads, z_off, n_rows

    @staticmethod
    def _out_hw(m, x):
        """Spatial size of module m's output for input view x."""
        h, w = x.H, x.W
        for sub in (m if isinstance(m, nn.Sequential) else [m]):
            if isinstance(sub, Upsample):
                h, w = 2 * h, 2 * w
            elif isinstance(sub, ShuffleV2Block):
                h, w = (h + 2 - 3) // sub.stride + 1, (w + 2 - 3) // sub.stride + 1
            elif isinstance(sub, Conv):
                h, w = (h + 2 * sub.p - sub.k) // sub.s + 1, (w + 2 * sub.p - sub.k) // sub.s + 1
            elif isinstance(sub, StemBlock):
                h, w = math.ceil(((h + 2 - 3) // 2 + 1) / 2), math.ceil(((w + 2 - 3) // 2 + 1) / 2)
        return h, w

    def _build(self, N, H, W, cache=None, frame_hw=None):
        pb, inp, heads, z_off, n_rows = self._emit(N, H, W, frame_hw)
        plan = CompiledPlan(pb, self._device(), cache)
        plan.input = plan.buf_tensor(inp, N) if inp is not None else None
        plan.frame_hw, plan.canvas_hw, plan.tables = frame_hw, (H, W), None
        plan.heads = [plan.buf_tensor(hb, N) for hb in heads]
        plan.z = plan.arena[z_off: z_off + N * n_rows * 16].view(N, n_rows, 16)
        plan.n_rows = n_rows
        return plan

    def plan_for(self, N, H=640, W=640, frame_hw=None):
        if self._device().type != "cuda":
            raise L.FacepathError("YOLOv5-face runs only on a HIP device (model.to('cuda')); there is no CPU path")
        key = (N, H, W, frame_hw, switch_key(PlanBuilder, Conv, StemBlock, C3, ShuffleV2Block, SPP, Concat, Model))
        return self._plans.get(key, lambda cache: self._build(N, H, W, cache, frame_hw))

    def run_plan(self, plan):
        """Forward + Detect decode on whatever is in plan.input.  Returns z (N, n_rows, 16): zero-copy, a view into
        the plan arena, valid until the next run of this plan."""
        plan.run()
        det = self.model[-1]
        lib = L.load()
        dev = self._device()
        row = 0
        ag = det.anchor_grid.view(det.nl, det.na, 2).cpu().numpy()
        for lvl, h in enumerate(plan.heads):
            N, ny, nx, _ = h.shape
            anc = (L.C.c_float * 6)(*[float(v) for v in ag[lvl].reshape(-1)])
            L.check(lib.fp_yolo_decode(L.ptr(h), N, ny, nx, det.na, float(det.stride[lvl]), anc, L.ptr(plan.z),
                                       plan.n_rows * 16, row, L.current_stream(dev)), "fp_yolo_decode")
            row += det.na * ny * nx
        return plan.z

    def forward(self, x, augment=False, profile=False):
        b, _, H, W = x.shape
        plan = self.plan_for(b, H, W)
        plan.input[..., :3].copy_(x.to(self._device(), torch.float32).permute(0, 2, 3, 1))
        plan.input[..., 3:].zero_()
        z = self.run_plan(plan).clone()              # plan.z / plan.heads are arena views the next call overwrites
        det = self.model[-1]
        heads = [h.view(b, h.shape[1], h.shape[2], det.na, det.no).permute(0, 3, 1, 2, 4).clone() for h in plan.heads]
        return z, heads


def parse_model(d, ch):
    """yolo.py:260-327 with names resolved from this module (fixes SURVEY F5)."""
    table = {"Conv": Conv, "StemBlock": StemBlock, "C3": C3, "ShuffleV2Block": ShuffleV2Block, "SPP": SPP,
             "Bottleneck": Bottleneck, "Concat": Concat, "nn.Upsample": Upsample, "Detect": Detect}
    anchors, nc, gd, gw = d["anchors"], d["nc"], d["depth_multiple"], d["width_multiple"]
    na = len(anchors[0]) // 2
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    ch = list(ch)
    for i, (f, n, mname, args) in enumerate(d["backbone"] + d["head"]):
        m = table[mname]
        args = [nc if a == "nc" else anchors if a == "anchors" else a for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        if m in (Conv, Bottleneck, SPP, C3, ShuffleV2Block, StemBlock):
            c1, c2 = ch[f], args[0]
            c2 = make_divisible(c2 * gw, 8) if c2 != no else c2
            args = [c1, c2, *args[1:]]
            if m is C3:
                args.insert(2, n)
                n = 1
        elif m is Concat:
            c2 = sum(ch[-1 if x == -1 else x + 1] for x in f)
        elif m is Detect:
            args.append([ch[x + 1] for x in f])
        elif m is Upsample:
            args = []
            c2 = ch[f]
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*[m(*args) for _ in range(n)]) if n > 1 else m(*args)
        m_.i, m_.f, m_.type = i, f, mname
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save), ch[1:]


def attempt_load(weights, map_location=None, cfg=None):
    """experimental.py:117-140.  The reference unpickles a whole Model object (weights_only=False); this loader
    only accepts data: a ``state_dict`` (or {'state_dict': ...} / {'model': state_dict}) read with
    ``weights_only=True``; the architecture comes from ``cfg`` or the file name (yolov5n / yolov5s / yolov5n-0.5)."""
    w = weights[0] if isinstance(weights, (list, tuple)) else weights
    ck = torch.load(w, map_location="cpu", weights_only=True)
    sd = ck.get("state_dict", ck.get("model", ck)) if isinstance(ck, dict) else ck
    if not isinstance(sd, dict):
        raise NotImplementedError("pickled Model checkpoints are not loaded (executes code); export its state_dict()")
    if cfg is None:
        base = os.path.basename(str(w)).lower()
        cfg = "yolov5n-0.5" if "n-0.5" in base or "n0.5" in base else "yolov5n" if "yolov5n" in base else "yolov5s"
    model = Model(cfg)
    model.load_state_dict(sd)
    dev = "cuda" if map_location is None else str(map_location).replace("hip", "cuda")
    return model.to(dev).fuse()

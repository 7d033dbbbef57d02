"""Mobile-FaceNet on MI355X: the reference's classes and ``state_dict`` keys
(face_detection_and_extraction/modules/mobile_facenet/mobile_facenet.py:39-154) with the forward pass
compiled to a HIP plan.  BatchNorm stays an epilogue affine (x*s + b), as the reference keeps it un-folded;
PReLU, the residual add and the final l2_norm are fused epilogues / one small kernel.
"""
import torch
import torch.nn as nn

from ... import _lib as L
from ...plan import CompiledPlan, PlanBuilder, PlanCache, bn_affine, switch_key
from ..params import BNParams, ConvParams, LinearParams, PReLUParams, _NoCompute, npy


def _affine(bn):
    return bn_affine(npy(bn.weight), npy(bn.bias), npy(bn.running_mean), npy(bn.running_var), bn.eps)


def l2_norm(input, axis=1):
    """mobile_facenet.py:30-33 — kept for API compatibility on host tensors that are already embeddings;
    inside the network the same formula is the FP_OP_L2NORM kernel."""
    raise RuntimeError("l2_norm runs inside the HIP plan (FP_OP_L2NORM); call MobileFaceNet.forward")


class Conv_block(_NoCompute):
    """conv (no bias) -> BatchNorm2d -> PReLU (mobile_facenet.py:39-51)."""

    def __init__(self, in_c, out_c, kernel=(1, 1), stride=(1, 1), padding=(0, 0), groups=1):
        super().__init__()
        self.k, self.s, self.p, self.groups = kernel[0], stride[0], padding[0], groups
        self.in_c, self.out_c = in_c, out_c
        self.conv = ConvParams(in_c, out_c, kernel[0], stride[0], padding[0], groups=groups, bias=False)
        self.bn = BNParams(out_c)
        self.prelu = PReLUParams(out_c)

    def emit(self, pb, x, act=True):
        OH = (x.H + 2 * self.p - self.k) // self.s + 1
        OW = (x.W + 2 * self.p - self.k) // self.s + 1
        y = pb.new_buf(OH, OW, self.out_c)
        s, b = _affine(self.bn)
        kw = dict(stride=self.s, pad=(self.p, self.p), scale=s, bias=b)
        if act:
            kw.update(slope=npy(self.prelu.weight), act=L.ACT_PRELU)
        if self.groups == 1:
            pb.conv(x, npy(self.conv.weight), y.view(), **kw)
        else:
            assert self.groups == self.in_c == self.out_c
            pb.dwconv(x, npy(self.conv.weight), y.view(), **kw)
        return y


class Linear_block(_NoCompute):
    """conv (no bias) -> BatchNorm2d (mobile_facenet.py:54-64)."""

    def __init__(self, in_c, out_c, kernel=(1, 1), stride=(1, 1), padding=(0, 0), groups=1):
        super().__init__()
        self.k, self.s, self.p, self.groups = kernel[0], stride[0], padding[0], groups
        self.in_c, self.out_c = in_c, out_c
        self.conv = ConvParams(in_c, out_c, kernel[0], stride[0], padding[0], groups=groups, bias=False)
        self.bn = BNParams(out_c)

    def emit(self, pb, x, res=None):
        OH = (x.H + 2 * self.p - self.k) // self.s + 1
        OW = (x.W + 2 * self.p - self.k) // self.s + 1
        y = pb.new_buf(OH, OW, self.out_c)
        s, b = _affine(self.bn)
        if self.groups == 1:
            pb.conv(x, npy(self.conv.weight), y.view(), stride=self.s, pad=(self.p, self.p), scale=s, bias=b,
                    res=res, res_mode=L.RES_ADD_AFTER_ACT if res is not None else L.RES_NONE)
        else:
            assert res is None
            pb.dwconv(x, npy(self.conv.weight), y.view(), stride=self.s, pad=(self.p, self.p), scale=s, bias=b)
        return y


class _X6Switch(type(_NoCompute)):
    """``Depth_Wise.X6`` reads / writes PlanBuilder.X6: one switch for every split-MFMA kernel."""

    @property
    def X6(cls):
        return PlanBuilder.X6

    @X6.setter
    def X6(cls, value):
        PlanBuilder.X6 = bool(value)


class Depth_Wise(_NoCompute, metaclass=_X6Switch):
    """1x1 expand (PReLU) -> depthwise 3x3 stride s (PReLU) -> 1x1 project (BN only) [+ x]
    (mobile_facenet.py:67-88)."""

    def __init__(self, in_c, out_c, residual=False, kernel=(3, 3), stride=(2, 2), padding=(1, 1), groups=1):
        super().__init__()
        self.conv = Conv_block(in_c, out_c=groups, kernel=(1, 1), padding=(0, 0), stride=(1, 1))
        self.conv_dw = Conv_block(groups, groups, groups=groups, kernel=kernel, padding=padding, stride=stride)
        self.project = Linear_block(groups, out_c, kernel=(1, 1), padding=(0, 0), stride=(1, 1))
        self.residual = residual

    FUSE = True   # class-wide switch: False emits the unfused DWCONV + CONV pair (A/B parity tests)
    # Whole block as ONE kernel (FP_OP_DWBLOCK, csrc/dwblock.hip) on the map sizes in BLOCK_SHAPES; None = block_policy
    # of the batch the plan will run on.  Tests pin it to a tuple.
    BLOCK_SHAPES = None
    # Depth_Wise.X6 (class attribute, = PlanBuilder.X6): every Depth_Wise block as ONE kernel on the bf16 matrix cores with
    # fp32-equivalent arithmetic (FP_OP_DWBLOCK + OPF_SPLIT3, csrc/dwblockx6.hip, csrc/split.h: exact three-way operand
    # split, six products), and the K >= 128 pointwise convs on csrc/pwx6.hip.  False = every GEMM on the fp32 MFMA (the
    # fmaf-chain kernels of rounds 1-3).

    @staticmethod
    def block_policy(n):
        """Map sizes on which the whole-block kernel beats the two-launch form at batch n, from whole-network A/B runs
        on MI355X (tools/mfn_probe.py, profiles/r03_mfn_probe.log): only the 7x7 blocks, when their tiles (three images
        each, one tile per CU at a time, ~66 us per round of 256 tiles) fill one round well: 33 us per forward at 528
        crops.  The 14x14 blocks never: a tile is a whole image, so ~528 crops need a third, nearly empty round (239 us
        against 184), and even at 1024 crops (4 full rounds) the six kernels run 320-377 us in the network (315 in
        isolation: the clock they reach after 2 ms of back-to-back launches) against 360 for the pair; 28x28 never
        (7-row bands recompute 2 of 9 expand rows)."""
        shapes = []
        if n >= 192 and -(-(-(-n // 3)) // 256) * 66.0 < 33.0 + 0.094 * n:
            shapes.append(7)
        return tuple(shapes)

    def emit(self, pb, x, expanded=None, in_dw=None):
        """expanded: the output of self.conv when the caller has already produced it (fused into the previous
        depthwise Conv_block, see MobileFaceNet._emit); x is then only the residual source.  in_dw: a depthwise
        Conv_block that precedes this block and is computed inside its kernel (FP_OPF_IN_DW: conv2_dw + conv_23)."""
        dw, pj, ex = self.conv_dw, self.project, self.conv
        shapes = Depth_Wise.BLOCK_SHAPES
        if shapes is None:
            shapes = getattr(pb, "dwblock_shapes", None)
        if shapes is None:
            shapes = Depth_Wise.block_policy(pb.N)
        if (Depth_Wise.FUSE and Depth_Wise.X6 and expanded is None and dw.k == 3 and dw.p == 1 and
                (ex.in_c, x.H) in pb.DWBLOCK_X6_SHAPES and pb.dwblock_supported(x, ex.in_c, ex.out_c, pj.out_c, dw.s)):
            y = pb.new_buf(x.H, x.W, pj.out_c)
            pb.dwblock(x, npy(ex.conv.weight), _affine(ex.bn), npy(ex.prelu.weight),
                       npy(dw.conv.weight), _affine(dw.bn), npy(dw.prelu.weight),
                       npy(pj.conv.weight), _affine(pj.bn), y.view(), self.residual, split=True)
            return y
        if (Depth_Wise.FUSE and Depth_Wise.X6 and expanded is None and dw.k == 3 and dw.p == 1 and not self.residual and
                pb.dwblock_x6d_supported(x, ex.in_c, ex.out_c, pj.out_c, dw.s)):
            y = pb.new_buf(x.H // 2, x.W // 2, pj.out_c)
            pb.dwblock(x, npy(ex.conv.weight), _affine(ex.bn), npy(ex.prelu.weight),
                       npy(dw.conv.weight), _affine(dw.bn), npy(dw.prelu.weight),
                       npy(pj.conv.weight), _affine(pj.bn), y.view(), False, split=True, stride=2,
                       in_dw=None if in_dw is None else (npy(in_dw.conv.weight), _affine(in_dw.bn), npy(in_dw.prelu.weight)))
            return y
        assert in_dw is None
        if (Depth_Wise.FUSE and expanded is None and x.H in shapes and dw.k == 3 and dw.p == 1 and
                pb.dwblock_supported(x, ex.in_c, ex.out_c, pj.out_c, dw.s)):
            y = pb.new_buf(x.H, x.W, pj.out_c)
            pb.dwblock(x, npy(ex.conv.weight), _affine(ex.bn), npy(ex.prelu.weight),
                       npy(dw.conv.weight), _affine(dw.bn), npy(dw.prelu.weight),
                       npy(pj.conv.weight), _affine(pj.bn), y.view(), self.residual)
            return y
        a = expanded if expanded is not None else self.conv.emit(pb, x)
        if (Depth_Wise.FUSE and dw.k == 3 and dw.p == 1 and dw.groups % 64 == 0 and pj.out_c % 4 == 0 and
                pj.out_c <= 128):
            OH = (a.H + 2 - 3) // dw.s + 1
            OW = (a.W + 2 - 3) // dw.s + 1
            y = pb.new_buf(OH, OW, pj.out_c)
            ds, db = _affine(dw.bn)
            ps, pbias = _affine(pj.bn)
            pb.dwpw(a.view(), npy(dw.conv.weight), ds, db, npy(dw.prelu.weight), npy(pj.conv.weight), ps, pbias,
                    y.view(), dw.s, res=x if self.residual else None)
            pb.free(a)
            return y
        b = self.conv_dw.emit(pb, a.view())
        pb.free(a)
        y = self.project.emit(pb, b.view(), res=x if self.residual else None)
        pb.free(b)
        return y


class Residual(_NoCompute):
    """num_block residual Depth_Wise blocks (mobile_facenet.py:91-101)."""

    def __init__(self, c, num_block, groups, kernel=(3, 3), stride=(1, 1), padding=(1, 1)):
        super().__init__()
        self.model = nn.Sequential(*[Depth_Wise(c, c, residual=True, kernel=kernel, padding=padding, stride=stride,
                                                groups=groups) for _ in range(num_block)])

    def emit(self, pb, xbuf):
        for blk in self.model:
            y = blk.emit(pb, xbuf.view())
            pb.free(xbuf)
            xbuf = y
        return xbuf


class Flatten(_NoCompute):
    pass


class MobileFaceNet(nn.Module):
    """mobile_facenet.py:104-154.  ``forward(x)``: (b, 3, 112, 112) float in [-1, 1] (BGR, as
    mobile_facenet/utils.py:13-17 feeds it) -> (b, embedding_size) unit-norm embeddings."""

    # conv2_dw + conv_23: True = depthwise launch + whole-block split-MFMA kernel; False = the round-2 pair of dw->pw kernels
    X6_CONV23 = True
    X6_CONV2_IN = True    # ... with conv2_dw inside that kernel's prologue (False: a separate depthwise launch)
    STEM_DW = True        # conv1 + conv2_dw as ONE kernel (FP_OPF_OUT_DW, csrc/stemdw.hip); conv_23 then takes conv2_dw's output

    def __init__(self, embedding_size):
        super().__init__()
        self.embedding_size = embedding_size
        self.conv1 = Conv_block(3, 64, kernel=(3, 3), stride=(2, 2), padding=(1, 1))
        self.conv2_dw = Conv_block(64, 64, kernel=(3, 3), stride=(1, 1), padding=(1, 1), groups=64)
        self.conv_23 = Depth_Wise(64, 64, kernel=(3, 3), stride=(2, 2), padding=(1, 1), groups=128)
        self.conv_3 = Residual(64, num_block=4, groups=128, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
        self.conv_34 = Depth_Wise(64, 128, kernel=(3, 3), stride=(2, 2), padding=(1, 1), groups=256)
        self.conv_4 = Residual(128, num_block=6, groups=256, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
        self.conv_45 = Depth_Wise(128, 128, kernel=(3, 3), stride=(2, 2), padding=(1, 1), groups=512)
        self.conv_5 = Residual(128, num_block=2, groups=256, kernel=(3, 3), stride=(1, 1), padding=(1, 1))
        self.conv_6_sep = Conv_block(128, 512, kernel=(1, 1), stride=(1, 1), padding=(0, 0))
        self.conv_6_dw = Linear_block(512, 512, groups=512, kernel=(7, 7), stride=(1, 1), padding=(0, 0))
        self.conv_6_flatten = Flatten()
        self.linear = LinearParams(512, embedding_size, bias=False)
        self.bn = BNParams(embedding_size)
        self._plans = PlanCache()

    def _device(self):
        return self.linear.weight.device

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._plans.clear()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._plans.clear()
        return out

    def _emit(self, N, H=112, W=112, block_shapes=None):
        """Emit the op list for batch N (host only, no GPU needed).  block_shapes: map sizes whose stride-1 Depth_Wise
        blocks become FP_OP_DWBLOCK (None: Depth_Wise.block_policy(N))."""
        pb = PlanBuilder(N)
        pb.dwblock_shapes = block_shapes
        inp = pb.new_buf(H, W, 3)
        c1, c2, c23 = self.conv1, self.conv2_dw, self.conv_23
        stem_dw = False
        if (Depth_Wise.FUSE and MobileFaceNet.STEM_DW and c1.groups == 1 and (c1.k, c1.s, c1.p) == (3, 2, 1) and
                (c2.k, c2.s, c2.p) == (3, 1, 1) and c2.groups == c2.in_c == c2.out_c == c1.out_c):
            y = pb.new_buf(H // 2, W // 2, c1.out_c)
            if pb.stem_dw_ok(inp.view(), npy(c1.conv.weight), y.view(), 2, (1, 1), L.ACT_PRELU, L.RES_NONE):
                # conv1 + conv2_dw in ONE kernel (FP_OPF_OUT_DW, csrc/stemdw.hip): conv1's 56 x 56 x 64 rows stay in LDS, the
                # depthwise conv runs on them there; conv_23 below then runs in its plain form (no FP_OPF_IN_DW prologue)
                s1, b1 = _affine(c1.bn)
                pb.conv(inp.view(), npy(c1.conv.weight), y.view(), stride=2, pad=(1, 1), scale=s1, bias=b1,
                        slope=npy(c1.prelu.weight), act=L.ACT_PRELU,
                        out_dw=(npy(c2.conv.weight), _affine(c2.bn), npy(c2.prelu.weight)))
                x, stem_dw = y, True
            else:
                pb.free(y)
        if not stem_dw:
            x = c1.emit(pb, inp.view())
        if stem_dw:
            y = c23.emit(pb, x.view()); pb.free(x); x = y
        elif (Depth_Wise.FUSE and Depth_Wise.X6 and MobileFaceNet.X6_CONV23 and c23.conv_dw.s == 2 and not c23.residual and
                x.H == x.W and c2.out_c == c23.conv.in_c and
                (c23.conv.in_c, c23.conv.out_c, c23.project.out_c, x.H) in pb.DWBLOCK_X6D_SHAPES):
            # conv2_dw + ALL of conv_23 (expand -> dw stride 2 -> project) as one split-MFMA kernel: conv2_dw is formed in
            # the kernel's prologue from an LDS image of conv1's rows (FP_OPF_IN_DW); neither its output (424 MB at 528
            # crops) nor the 128-channel 56x56 tensor (848 MB) ever exists
            if MobileFaceNet.X6_CONV2_IN and c2.k == 3 and c2.s == 1 and c2.p == 1 and c2.groups == c2.in_c == c2.out_c:
                y = c23.emit(pb, x.view(), in_dw=c2); pb.free(x); x = y
            else:
                y = c2.emit(pb, x.view()); pb.free(x); x = y
                y = c23.emit(pb, x.view()); pb.free(x); x = y
        elif Depth_Wise.FUSE and c2.k == 3 and c2.s == 1 and c2.p == 1 and c2.groups % 64 == 0 and not c23.residual:
            # conv2_dw (dw3x3 + BN + PReLU) -> conv_23.conv (1x1 + BN + PReLU) as ONE dw->pw kernel: the 64-channel
            # 56x56 tensor between them (873 MB at N = 1088) never goes to HBM
            ex = c23.conv
            a = pb.new_buf(x.H, x.W, ex.out_c)
            ds, db = _affine(c2.bn)
            es, eb = _affine(ex.bn)
            pb.dwpw(x.view(), npy(c2.conv.weight), ds, db, npy(c2.prelu.weight), npy(ex.conv.weight), es, eb,
                    a.view(), 1, out_slope=npy(ex.prelu.weight))
            pb.free(x)
            x = c23.emit(pb, None, expanded=a)
        else:
            y = c2.emit(pb, x.view()); pb.free(x); x = y
            y = c23.emit(pb, x.view()); pb.free(x); x = y
        x = self.conv_3.emit(pb, x)
        y = self.conv_34.emit(pb, x.view()); pb.free(x); x = y
        x = self.conv_4.emit(pb, x)
        y = self.conv_45.emit(pb, x.view()); pb.free(x); x = y
        x = self.conv_5.emit(pb, x)
        y = self.conv_6_sep.emit(pb, x.view()); pb.free(x); x = y
        y = self.conv_6_dw.emit(pb, x.view()); pb.free(x); x = y        # (N, 1, 1, 512)
        assert (x.H, x.W) == (1, 1), "Mobile-FaceNet expects 112x112 inputs (7x7 map before conv_6_dw)"
        # Linear(512, E, bias=False) + BatchNorm1d as one 1x1 conv with an affine epilogue, then l2_norm
        E = self.embedding_size
        z = pb.new_buf(1, 1, E)
        s, b = _affine(self.bn)
        pb.conv(x.view(), npy(self.linear.weight).reshape(E, 512, 1, 1), z.view(0, E), scale=s, bias=b)
        o = pb.new_buf(1, 1, E)
        pb.l2norm(z.view(0, E), o.view(0, E))
        return pb, inp, o

    def _build(self, N, cache=None, block_shapes=None):
        E = self.embedding_size
        pb, inp, o = self._emit(N, block_shapes=block_shapes)
        plan = CompiledPlan(pb, self._device(), cache)
        plan.input = plan.buf_tensor(inp, N)
        plan.out = plan.buf_tensor(o, N).view(N, -1)[:, :E]
        return plan

    def plan_for(self, N, n_run=None):
        """The plan with batch capacity N.  n_run: the batch it is about to run on (a prefix of its capacity,
        CompiledPlan.run(n)); which blocks use the whole-block kernel follows n_run (Depth_Wise.block_policy), so a
        capacity can have up to four plans (the 14x14 / 7x7 choices), each with its own arena."""
        if self._device().type != "cuda":
            raise L.FacepathError("MobileFaceNet runs only on a HIP device (model.to('cuda')); there is no CPU path")
        shapes = Depth_Wise.BLOCK_SHAPES if Depth_Wise.BLOCK_SHAPES is not None else \
            Depth_Wise.block_policy(N if n_run is None else n_run)
        shapes = tuple(shapes)
        if Depth_Wise.FUSE and Depth_Wise.X6:
            # every stride-1 block takes the split whole-block kernel before `shapes` is consulted (Depth_Wise.emit): the op
            # list does not depend on it, so ONE capacity = ONE plan and one arena whatever n_run is
            shapes = tuple(h for h in shapes if (128 if h < 28 else 64, h) not in PlanBuilder.DWBLOCK_X6_SHAPES)
        key = (N, shapes, switch_key(PlanBuilder, Depth_Wise, MobileFaceNet))
        return self._plans.get(key, lambda cache: self._build(N, cache, block_shapes=shapes))

    def forward(self, x):
        b = x.shape[0]
        plan = self.plan_for(b)
        plan.input[..., :3].copy_(x.to(self._device(), torch.float32).permute(0, 2, 3, 1))
        plan.input[..., 3:].zero_()
        plan.run()
        return plan.out.clone()                      # plan.out is an arena view the next call overwrites

    def embed_resident(self, n):
        """Run the plan on whatever fp_resize_normalize wrote into plan_for(n).input; returns (n, E).
        Zero-copy: the result is a view into the plan arena, valid until the next run at this batch size."""
        plan = self.plan_for(n)
        plan.run()
        return plan.out

"""BlazeFace on MI355X: same classes, attributes and ``state_dict`` keys as the reference
(face_detection_and_extraction/modules/blazeface/blazeface.py), with the arithmetic in HIP:

  forward (blazeface.py:192-228)                 -> fp_plan_run   (csrc/conv.hip, csrc/blaze.hip)
  _tensors_to_detections/_decode_boxes (:321-402) -> fp_blaze_decode        (csrc/post.hip)
  _weighted_non_max_suppression (:404-458)        -> fp_blaze_weighted_nms  (csrc/post.hip)
  _preprocess x/127.5-1 (:248-250)                -> LUT inside fp_resize_normalize (csrc/image.hip)
"""
import os

import numpy as np
import torch
import torch.nn as nn

from ... import _lib as L
from ...plan import Buf, CompiledPlan, PlanBuilder, PlanCache, View, cpad, switch_key
from ..params import ConvParams, _NoCompute, npy


class BlazeBlock(_NoCompute):
    """blazeface.py:12-47.  convs = [depthwise kxk (stride s), pointwise 1x1]; shortcut = x or
    maxpool2(x), zero-padded on channels; ReLU(convs(h) + shortcut)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super().__init__()
        self.stride = stride
        self.channel_pad = out_channels - in_channels
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        padding = 0 if stride == 2 else (kernel_size - 1) // 2
        self.convs = nn.Sequential(
            ConvParams(in_channels, in_channels, kernel_size, stride, padding, groups=in_channels, bias=True),
            ConvParams(in_channels, out_channels, 1, 1, 0, bias=True))

    FUSE = True   # class-wide switch: False emits the unfused DWCONV + CONV pair (A/B parity tests)
    ROWPAD = True  # class-wide switch: False keeps every activation dense (A/B parity tests)
    PAIR = os.environ.get("FP_BLAZE_PAIR", "1") == "1"    # class-wide switch: two consecutive stride-1 24 -> 24 blocks as ONE op (FP_OP_BLAZEPAIR, csrc/blazepair.hip)

    CHAIN = os.environ.get("FP_BLAZE_CHAIN", "1") == "1"   # class-wide switch: a run of stride-1 96 -> 96 blocks on the 16 x 16 map as ONE op (FP_OP_BLAZECHAIN, csrc/blazechain.hip)

    def chains(self):
        """True if this block can be a member of an FP_OP_BLAZECHAIN run."""
        return self.kernel_size == 3 and self.stride == 1 and self.in_channels == 96 and self.out_channels == 96

    def pairs_with(self, other, pb, x):
        """True if self followed by `other`, fed the row-padded view x, runs as one FP_OP_BLAZEPAIR."""
        ok = lambda b: (isinstance(b, BlazeBlock) and b.kernel_size == 3 and b.stride == 1 and b.in_channels == 24 and
                        b.out_channels == 24)
        return (BlazeBlock.FUSE and BlazeBlock.ROWPAD and BlazeBlock.PAIR and ok(self) and ok(other) and
                pb.blazepair_supported(x))

    PAIR_S2 = os.environ.get("FP_BLAZE_PAIR_S2", "1") == "1"   # class-wide switch: the single stride-1 24 -> 24 block that ends a stage and
                                                                # the stride-2 block behind it as ONE op (csrc/blazepairs2.hip)

    def pairs_with_s2(self, other, pb, x):
        """True if self (stride 1, 24 -> 24) followed by the stride-2 block `other`, fed the row-padded view x, runs as one
        FP_OP_BLAZEPAIR with stride = 2."""
        return (BlazeBlock.FUSE and BlazeBlock.ROWPAD and BlazeBlock.PAIR_S2 and isinstance(other, BlazeBlock) and
                self.kernel_size == 3 and self.stride == 1 and self.in_channels == 24 and self.out_channels == 24 and
                other.kernel_size == 3 and other.stride == 2 and other.in_channels == 24 and
                pb.blazepair_s2_supported(x, other.out_channels))

    def emit_pair_s2(self, other, pb, x, out_rowpad=False):
        y = (pb.new_buf_rowpad if out_rowpad else pb.new_buf)(x.H // 2, x.W // 2, other.out_channels)
        pb.blazepair_s2(x, [(npy(b.convs[0].weight), npy(b.convs[0].bias), npy(b.convs[1].weight), npy(b.convs[1].bias))
                            for b in (self, other)], y.view())
        return y

    def emit_pair(self, other, pb, x, out_rowpad=False):
        y = (pb.new_buf_rowpad if out_rowpad else pb.new_buf)(x.H, x.W, 24)
        pb.blazepair(x, [(npy(b.convs[0].weight), npy(b.convs[0].bias), npy(b.convs[1].weight), npy(b.convs[1].bias))
                         for b in (self, other)], y.view())
        return y

    def wide_ok(self, H, W):
        """True if this block on an H x W map runs on the small-map wave-private kernel (csrc/blazewp.hip
        fp_blazeblock_wps_eligible: stride 1, 48 -> 48 or 96 -> 96, 16- or 32-pixel-wide maps, row-padded input)."""
        return (BlazeBlock.FUSE and BlazeBlock.ROWPAD and self.kernel_size == 3 and self.stride == 1 and
                self.in_channels == self.out_channels and self.in_channels in (48, 96) and W in (16, 32) and
                (H * W) % 32 == 0 and H * W >= 64)

    def fused(self, pb, x):
        """True if emit() takes the fused FP_OP_BLAZEBLOCK path for input view x."""
        if self.wide_ok(x.H, x.W) and x.coff == 0:
            return True
        OW = x.W // 2 if self.stride == 2 else x.W
        return (BlazeBlock.FUSE and self.kernel_size == 3 and x.coff == 0 and OW % 4 == 0 and
                pb.blazeblock_lds_bytes(x.C, cpad(self.out_channels)) <= 80 * 1024)

    def wants_rowpad_input(self, H, W):
        """True if this block, fed an H x W map, runs on a wave-private kernel that reads a row-padded input
        (csrc/blazewp.hip fp_blazeblock_wp_eligible: stride 1, 24 -> 24, rows of whole 32-pixel tiles; or wide_ok)."""
        if self.wide_ok(H, W):
            return True
        return (BlazeBlock.FUSE and BlazeBlock.ROWPAD and self.kernel_size == 3 and self.stride == 1 and
                self.in_channels == 24 and self.out_channels == 24 and W % 32 == 0 and W >= 64 and H % 4 == 0 and H >= 8)

    def emit(self, pb, x, out_rowpad=False):
        """out_rowpad: write the output in the row-padded layout (the next block asked for it); only the fused path
        can."""
        dw, pw = self.convs[0], self.convs[1]
        if self.wide_ok(x.H, x.W) and x.coff == 0:
            tmp = None
            if not x.buf.rowpad:   # dense producer (the unfused stride-2 block before a 96-channel stage): one copy
                tmp = pb.new_buf_rowpad(x.H, x.W, x.C)
                pb.copy(x, tmp.view())
                x = tmp.view()
            y = (pb.new_buf_rowpad if out_rowpad else pb.new_buf)(x.H, x.W, self.out_channels)
            pb.blazeblock(x, npy(dw.weight), npy(dw.bias), npy(pw.weight), npy(pw.bias), y.view(), 1)
            if tmp is not None:
                pb.free(tmp)
            return y
        if self.stride == 2:
            # h = F.pad(x, (0, 2, 0, 2)); dw stride 2, padding 0 (blazeface.py:38-39)
            OH, OW, pad = x.H // 2, x.W // 2, (0, 0)
            res_mode = L.RES_POOL2_BEFORE_ACT
        else:
            OH, OW, pad = x.H, x.W, (1, 1)
            res_mode = L.RES_ADD_BEFORE_ACT
        if self.fused(pb, x):
            y = (pb.new_buf_rowpad if out_rowpad else pb.new_buf)(OH, OW, self.out_channels)
            pb.blazeblock(x, npy(dw.weight), npy(dw.bias), npy(pw.weight), npy(pw.bias), y.view(), self.stride)
            return y
        assert not out_rowpad and not x.buf.rowpad
        t = pb.new_buf(OH, OW, self.in_channels)
        pb.dwconv(x, npy(dw.weight), t.view(), stride=self.stride, pad=pad, bias=npy(dw.bias))
        y = pb.new_buf(OH, OW, self.out_channels)
        res = View(x.buf, x.coff, min(self.in_channels, x.C))
        pb.conv(t.view(), npy(pw.weight), y.view(), bias=npy(pw.bias), act=L.ACT_RELU, res=res, res_mode=res_mode)
        pb.free(t)
        return y


class FinalBlazeBlock(_NoCompute):
    """blazeface.py:50-68: pad (0,2,0,2), dw 3x3 s2, 1x1, ReLU, no shortcut."""

    def __init__(self, channels, kernel_size=3):
        super().__init__()
        self.channels = channels
        self.convs = nn.Sequential(
            ConvParams(channels, channels, kernel_size, 2, 0, groups=channels, bias=True),
            ConvParams(channels, channels, 1, 1, 0, bias=True))

    def emit(self, pb, x):
        dw, pw = self.convs[0], self.convs[1]
        OH, OW = x.H // 2, x.W // 2
        t = pb.new_buf(OH, OW, self.channels)
        pb.dwconv(x, npy(dw.weight), t.view(), stride=2, pad=(0, 0), bias=npy(dw.bias))
        y = pb.new_buf(OH, OW, self.channels)
        pb.conv(t.view(), npy(pw.weight), y.view(), bias=npy(pw.bias), act=L.ACT_RELU)
        pb.free(t)
        return y


class _ReLUTag(_NoCompute):
    """Placeholder for nn.ReLU in the backbone Sequential so child indices (state_dict keys) match."""


def generate_anchors(back_model=False):
    """MediaPipe SSD anchors for BlazeFace (the reference loads them from anchors.npy / anchorsback.npy,
    blazeface.py:238-246, files that do not ship in the tree).  896 rows (x_center, y_center, 1, 1):
    a 16x16 grid with 2 anchors per cell, then an 8x8 grid with 6 (fixed_anchor_size, aspect ratio 1)."""
    rows = []
    for grid, per_cell in ((16, 2), (8, 6)):
        for y in range(grid):
            for x in range(grid):
                for _ in range(per_cell):
                    rows.append([(x + 0.5) / grid, (y + 0.5) / grid, 1.0, 1.0])
    return np.asarray(rows, dtype=np.float32)


class BlazeFace(nn.Module):
    """blazeface.py:71-458.  ``forward(x)`` takes the pre-processed NCHW float batch and returns ``[r, c]``
    with r (b, 896, 16) and c (b, 896, 1); ``predict_on_batch`` returns a list of (k, 17) tensors
    (ymin, xmin, ymax, xmax, 6 keypoints, score)."""

    # class-wide switch: the back model's 5 x 5 stem on u8 frames on the bf16 matrix cores (FP_OP_STEM_U8 + FP_OPF_SPLIT3,
    # stem5_u8_x6_kernel; needs PlanBuilder.X6 as every split kernel does).  False: the fp32-MFMA band kernel, whose results are
    # bit-identical to the stand-alone letterbox + conv (the tests of that identity switch it off)
    STEM_X6 = os.environ.get("FP_BLAZE_STEM_X6", "1") == "1"

    def __init__(self, back_model=False):
        super().__init__()
        self.num_classes = 1
        self.num_anchors = 896
        self.num_coords = 16
        self.score_clipping_thresh = 100.0
        self.back_model = back_model
        if back_model:
            self.x_scale = self.y_scale = self.h_scale = self.w_scale = 256.0
            self.min_score_thresh = 0.65
        else:
            self.x_scale = self.y_scale = self.h_scale = self.w_scale = 128.0
            self.min_score_thresh = 0.75
        self.min_suppression_threshold = 0.3
        self.anchors = None
        self._plans = PlanCache()
        # True: this network's plans run BESIDE another network's kernels (FacePipeline(two_streams=True)).  Ops whose
        # workgroups own a whole CU are then not emitted: FP_OP_BLAZECHAIN holds 157 KB of LDS, so for its 75 us nothing of
        # the other stream fits on the CUs and its own launch waits for them to drain -- measured 4.10 against 4.01 ms
        # per two-stream step, although the detector alone gets 0.19 ms faster with it (FINDINGS.md finding 30).
        self.co_scheduled = False
        self._define_layers()

    def _define_layers(self):
        stem = [ConvParams(3, 24, 5, 2, 0, bias=True), _ReLUTag()]
        if self.back_model:
            spec = ([(24, 24, 1)] * 7 + [(24, 24, 2)] + [(24, 24, 1)] * 7 + [(24, 48, 2)] + [(48, 48, 1)] * 7 +
                    [(48, 96, 2)] + [(96, 96, 1)] * 7)
            self.backbone = nn.Sequential(*stem, *[BlazeBlock(i, o, stride=s) for i, o, s in spec])
            self.final = FinalBlazeBlock(96)
            c8 = 96
        else:
            spec1 = [(24, 24, 1), (24, 28, 1), (28, 32, 2), (32, 36, 1), (36, 42, 1), (42, 48, 2), (48, 56, 1),
                     (56, 64, 1), (64, 72, 1), (72, 80, 1), (80, 88, 1)]
            spec2 = [(88, 96, 2)] + [(96, 96, 1)] * 4
            self.backbone1 = nn.Sequential(*stem, *[BlazeBlock(i, o, stride=s) for i, o, s in spec1])
            self.backbone2 = nn.Sequential(*[BlazeBlock(i, o, stride=s) for i, o, s in spec2])
            c8 = 88
        self.classifier_8 = ConvParams(c8, 2, 1, bias=True)
        self.classifier_16 = ConvParams(96, 6, 1, bias=True)
        self.regressor_8 = ConvParams(c8, 32, 1, bias=True)
        self.regressor_16 = ConvParams(96, 96, 1, bias=True)

    # ------------------------------------------------------------------ loading
    def _device(self):
        return self.classifier_8.weight.device

    def load_weights(self, path):
        self.load_state_dict(torch.load(path, weights_only=True))
        self.eval()

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._plans.clear()
        return out

    def load_anchors(self, path, use_numpy=False):
        arr = np.load(path).astype(np.float32)
        self.set_anchors(arr)

    def set_anchors(self, arr):
        arr = np.asarray(arr, dtype=np.float32)
        assert arr.ndim == 2 and arr.shape[0] == self.num_anchors and arr.shape[1] == 4
        self.anchors = torch.tensor(arr, dtype=torch.float32, device=self._device())

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._plans.clear()
        if self.anchors is not None:
            self.anchors = fn(self.anchors)
        return out

    # ------------------------------------------------------------------ plan
    @property
    def input_hw(self):
        return (256, 256) if self.back_model else (128, 128)

    FUSE_LETTERBOX = True    # class-wide switch: False keeps the stand-alone letterbox kernel (A/B parity tests)

    def _emit(self, N, frame_hw=None):
        """Emit the op list for batch N (host only, no GPU needed).  frame_hw = (frame_h, frame_w): the stem reads u8
        frames of that size itself (FP_OP_STEM_U8: no fp32 canvas, no letterbox launch; external buffers 0..2 =
        frames, tap tables, LUT); None: the plan input is the NHWC fp32 canvas."""
        H, W = self.input_hw
        pb = PlanBuilder(N)
        inp = pb.new_buf(H, W, 3) if frame_hw is None else None      # NHWC, channel-padded to 4
        seq = list(self.backbone) if self.back_model else list(self.backbone1)
        stem = seq[0]
        # A tensor is kept in the row-padded layout (include/facepath.h) when its consumer is a 24 -> 24 stride-1 block
        # on a map of whole 32-pixel rows (the wave-private kernel) and its producer can write it (the stem kernel, a
        # fused block).
        blocks = seq[2:]

        def chain_len(i, h, w):    # blocks[i:i + n] on an h x w map run as one FP_OP_BLAZECHAIN (0: they do not)
            if not (BlazeBlock.FUSE and BlazeBlock.CHAIN and (not self.co_scheduled or os.environ.get("FP_CHAIN_CO") == "1") and
                    PlanBuilder.X6 and (h, w) == (16, 16)):
                return 0
            n = 0
            while i + n < len(blocks) and isinstance(blocks[i + n], BlazeBlock) and blocks[i + n].chains():
                n += 1
            return min(n, 16) if n >= 2 else 0

        def rowpad_for(i, h, w):   # should the input of blocks[i] (an h x w map) be row-padded?
            if chain_len(i, h, w):     # the chain kernel reads a dense map
                return False
            return i < len(blocks) and isinstance(blocks[i], BlazeBlock) and blocks[i].wants_rowpad_input(h, w)

        # F.pad(x, (1, 2, 1, 2)) + 5x5 stride-2 conv + ReLU (blazeface.py:118-120,195)
        x = (pb.new_buf_rowpad if rowpad_for(0, H // 2, W // 2) else pb.new_buf)(H // 2, W // 2, 24)
        if frame_hw is None:
            pb.conv(inp.view(), npy(stem.weight), x.view(), stride=2, pad=(1, 1), bias=npy(stem.bias), act=L.ACT_RELU)
        else:
            split = BlazeFace.STEM_X6 and PlanBuilder.X6 and (H, W) == (256, 256) and N >= 16
            pb.stem_u8((H, W, frame_hw[0], frame_hw[1], 0), npy(stem.weight), x.view(), pad=(1, 1), bias=npy(stem.bias),
                       act=L.ACT_RELU, split=split)
        i = 0
        while i < len(blocks):
            blk = blocks[i]
            nchain = chain_len(i, x.H, x.W)
            if nchain and pb.blazechain_supported(x.view()):
                y = pb.new_buf(16, 16, 96)
                pb.blazechain(x.view(), [(npy(b.convs[0].weight), npy(b.convs[0].bias), npy(b.convs[1].weight), npy(b.convs[1].bias))
                                         for b in blocks[i:i + nchain]], y.view())
                i += nchain - 1
            elif (isinstance(blk, BlazeBlock) and i + 1 < len(blocks) and isinstance(blocks[i + 1], BlazeBlock) and
                    blk.pairs_with(blocks[i + 1], pb, x.view())):
                y = blk.emit_pair(blocks[i + 1], pb, x.view(), out_rowpad=rowpad_for(i + 2, x.H, x.W))
                i += 1
            elif isinstance(blk, BlazeBlock) and i + 1 < len(blocks) and blk.pairs_with_s2(blocks[i + 1], pb, x.view()):
                y = blk.emit_pair_s2(blocks[i + 1], pb, x.view(), out_rowpad=rowpad_for(i + 2, x.H // 2, x.W // 2))
                i += 1
            elif isinstance(blk, BlazeBlock):
                oh, ow = (x.H // 2, x.W // 2) if blk.stride == 2 else (x.H, x.W)
                y = blk.emit(pb, x.view(), out_rowpad=blk.fused(pb, x.view()) and rowpad_for(i + 1, oh, ow))
            else:
                y = blk.emit(pb, x.view())
            pb.free(x)
            x = y
            i += 1
        if self.back_model:
            h = self.final.emit(pb, x.view())
        else:
            h = x
            for blk in self.backbone2:
                y = blk.emit(pb, h.view())
                if h is not x:
                    pb.free(h)
                h = y
        # heads: 1x1 convs written straight into the concatenated (b, 896, 16) / (b, 896, 1) tensors
        # (blazeface.py:209-228: NHWC permute + reshape + cat are pure addressing here)
        A = self.num_anchors
        r_off, r_size = pb.new_raw(A * 16)
        c_off, c_size = pb.new_raw(A)
        def head(src, conv, off, per_img, skip_rows, row_floats):
            cout = conv.weight.shape[0]
            hb = Buf(src.H, src.W, cout, off + skip_rows * row_floats, 0, ns_=per_img)
            pb.conv(src.view(), npy(conv.weight), View(hb, 0, cout), bias=npy(conv.bias))
        head(x, self.classifier_8, c_off, A, 0, 1)
        head(h, self.classifier_16, c_off, A, 512, 1)
        head(x, self.regressor_8, r_off, A * 16, 0, 16)
        head(h, self.regressor_16, r_off, A * 16, 512, 16)
        return pb, inp, r_off, c_off

    def _build(self, N, cache=None, frame_hw=None):
        A = self.num_anchors
        pb, inp, r_off, c_off = self._emit(N, frame_hw)
        plan = CompiledPlan(pb, self._device(), cache)
        plan.inp = inp
        plan.r = plan.arena[r_off: r_off + N * A * 16].view(N, A, 16)
        plan.c = plan.arena[c_off: c_off + N * A].view(N, A, 1)
        plan.input = plan.buf_tensor(inp, N) if inp is not None else None
        plan.frame_hw, plan.canvas_hw, plan.tables = frame_hw, self.input_hw, None
        return plan

    def plan_for(self, N, frame_hw=None):
        if self._device().type != "cuda":
            raise L.FacepathError("BlazeFace runs only on a HIP device (model.to('cuda')); there is no CPU path")
        key = (N, frame_hw, self.co_scheduled, os.environ.get("FP_CHAIN_CO") == "1",
               switch_key(PlanBuilder, BlazeBlock, BlazeFace))
        return self._plans.get(key, lambda cache: self._build(N, cache, frame_hw))

    # ------------------------------------------------------------------ inference
    def forward(self, x):
        """x: (b, 3, H, W) float, already in [-1, 1] (blazeface.py:192-228)."""
        b = x.shape[0]
        plan = self.plan_for(b)
        plan.input[..., :3].copy_(x.to(self._device(), torch.float32).permute(0, 2, 3, 1))
        plan.input[..., 3:].zero_()
        plan.run()
        return [plan.r.clone(), plan.c.clone()]     # plan.r / plan.c are arena views the next call overwrites

    def _preprocess_lut(self):
        # x.float() / 127.5 - 1.0 (blazeface.py:248-250), evaluated for the 256 u8 values in torch fp32
        if getattr(self, "_lut", None) is None or self._lut.device != self._device():
            self._lut = (torch.arange(256, dtype=torch.float32) / 127.5 - 1.0).to(self._device())
        return self._lut

    def predict_on_image(self, img):
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(img.copy()).permute((2, 0, 1))
        return self.predict_on_batch(img.unsqueeze(0))[0]

    def raw_from_u8_nhwc(self, frames_u8):
        """(b, H, W, 3) u8 RGB on device, H/W = the model input size -> raw (r, c) of the HIP plan.
        Zero-copy: r and c are views into the plan arena, valid until the next run at this batch size."""
        b, H, W, _ = frames_u8.shape
        assert (H, W) == self.input_hw
        plan = self.plan_for(b)
        lib = L.load()
        items = torch.tensor([[i, 0, 0, W, H, 0, 0, W, H] for i in range(b)], dtype=torch.int32,
                             device=self._device())
        L.check(lib.fp_resize_normalize(L.ptr(frames_u8), b, H, W, L.ptr(items), b, L.ptr(plan.input), H, W,
                                        plan.input.shape[-1], L.ptr(self._preprocess_lut()), 0, 0,
                                        L.current_stream(self._device())), "fp_resize_normalize")
        plan.run()
        return plan.r, plan.c

    def predict_on_batch(self, x, use_numpy_for_post_proc=False):
        """blazeface.py:266-319.  x: (b, H, W, 3) numpy u8 or (b, 3, H, W) tensor, RGB, model-input sized."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x).permute((0, 3, 1, 2))
        assert x.shape[1] == 3
        assert x.shape[2] == self.input_hw[0] and x.shape[3] == self.input_hw[1]
        dev = self._device()
        if x.dtype == torch.uint8:
            r, c = self.raw_from_u8_nhwc(x.to(dev).permute(0, 2, 3, 1).contiguous())
        else:
            r, c = self.forward(x.to(dev).float() / 127.5 - 1.0)
        dets, counts = self.postprocess(r, c)
        counts = counts.cpu().tolist()
        out = []
        for i, k in enumerate(counts):
            faces = dets[i, :k].clone() if k > 0 else torch.zeros((0, 17), device=dev)
            out.append(faces.cpu().numpy() if use_numpy_for_post_proc else faces)
        return out

    # ---- the reference's private post-processing names (blazeface.py:321, :373, :404), same arguments and return
    # shapes, on the device kernels.  A caller that used them on the reference's BlazeFace drops in.
    def _as_dev(self, t):
        return torch.as_tensor(np.asarray(t) if isinstance(t, np.ndarray) else t).to(self._device(), torch.float32).contiguous()

    def _decode_boxes(self, raw_boxes, anchors, use_numpy=False):
        """blazeface.py:373-402: anchor decode of the whole batch, (b, 896, 16) -> (b, 896, 16) [ymin, xmin, ymax, xmax,
        6 keypoints].  fp_blaze_decode with a threshold nothing fails, so every anchor comes out, in anchor order."""
        rb, an = self._as_dev(raw_boxes), self._as_dev(anchors)
        b, A = rb.shape[0], rb.shape[1]
        dev = self._device()
        cand = torch.empty((b, A, 17), dtype=torch.float32, device=dev)
        cnt = torch.empty((b,), dtype=torch.int32, device=dev)
        ones = torch.ones((b, A), dtype=torch.float32, device=dev)
        L.check(L.load().fp_blaze_decode(L.ptr(rb), L.ptr(ones), L.ptr(an), b, A, self.x_scale, self.y_scale, self.w_scale,
                                         self.h_scale, self.score_clipping_thresh, -1.0, L.ptr(cand), L.ptr(cnt),
                                         L.current_stream(dev)), "fp_blaze_decode")
        boxes = cand[..., :16].contiguous()
        return boxes.cpu().numpy() if use_numpy else boxes

    def _tensors_to_detections(self, raw_box_tensor, raw_score_tensor, anchors, use_numpy=False):
        """blazeface.py:321-371: raw (b, 896, 16) + (b, 896, 1) -> a list of (num_detections, 17) per image (decode,
        clip, sigmoid, score >= min_score_thresh), in anchor order."""
        assert raw_box_tensor.ndim == 3
        assert raw_box_tensor.shape[1] == self.num_anchors
        assert raw_box_tensor.shape[2] == self.num_coords
        assert raw_score_tensor.ndim == 3
        assert raw_score_tensor.shape[1] == self.num_anchors
        assert raw_score_tensor.shape[2] == self.num_classes
        assert raw_box_tensor.shape[0] == raw_score_tensor.shape[0]
        rb, rs, an = self._as_dev(raw_box_tensor), self._as_dev(raw_score_tensor), self._as_dev(anchors)
        b, A = rb.shape[0], self.num_anchors
        dev = self._device()
        cand = torch.empty((b, A, 17), dtype=torch.float32, device=dev)
        cnt = torch.empty((b,), dtype=torch.int32, device=dev)
        L.check(L.load().fp_blaze_decode(L.ptr(rb), L.ptr(rs), L.ptr(an), b, A, self.x_scale, self.y_scale, self.w_scale,
                                         self.h_scale, self.score_clipping_thresh, self.min_score_thresh, L.ptr(cand),
                                         L.ptr(cnt), L.current_stream(dev)), "fp_blaze_decode")
        out = []
        for i, k in enumerate(cnt.cpu().tolist()):
            d = cand[i, :k].clone()
            out.append(d.cpu().numpy() if use_numpy else d)
        return out

    def _weighted_non_max_suppression(self, detections, use_numpy=False):
        """blazeface.py:404-458: one image's (count, 17) detections -> a list of blended (17,) detections."""
        if len(detections) == 0:
            return []
        d = self._as_dev(detections)
        n = d.shape[0]
        dev = self._device()
        cnt = torch.tensor([n], dtype=torch.int32, device=dev)
        out = torch.empty((1, n, 17), dtype=torch.float32, device=dev)
        oc = torch.empty((1,), dtype=torch.int32, device=dev)
        L.check(L.load().fp_blaze_weighted_nms(L.ptr(d), L.ptr(cnt), 1, n, self.min_suppression_threshold, L.ptr(out),
                                               L.ptr(oc), None, L.current_stream(dev)), "fp_blaze_weighted_nms")
        k = int(oc.item())
        rows = out[0, :k]
        return [r.cpu().numpy() for r in rows] if use_numpy else [r.clone() for r in rows]

    def postprocess(self, r, c):
        """decode + threshold + weighted NMS on device.  Returns (dets [b, 896, 17], counts [b])."""
        assert self.anchors is not None, "call load_anchors()/set_anchors() first"
        lib = L.load()
        b, A = r.shape[0], self.num_anchors
        dev = self._device()
        s = L.current_stream(dev)
        cand = torch.empty((b, A, 17), dtype=torch.float32, device=dev)
        ccount = torch.empty((b,), dtype=torch.int32, device=dev)
        L.check(lib.fp_blaze_decode(L.ptr(r), L.ptr(c), L.ptr(self.anchors), b, A, self.x_scale, self.y_scale,
                                    self.w_scale, self.h_scale, self.score_clipping_thresh, self.min_score_thresh,
                                    L.ptr(cand), L.ptr(ccount), s), "fp_blaze_decode")
        out = torch.empty((b, A, 17), dtype=torch.float32, device=dev)
        ocount = torch.empty((b,), dtype=torch.int32, device=dev)
        L.check(lib.fp_blaze_weighted_nms(L.ptr(cand), L.ptr(ccount), b, A, self.min_suppression_threshold,
                                          L.ptr(out), L.ptr(ocount), None, s), "fp_blaze_weighted_nms")
        self._last_candidates = (cand, ccount)
        return out, ocount

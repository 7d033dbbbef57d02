"""Plan builder: turns a network description into the flat ``fp_op`` array that
``fp_plan_run`` (include/facepath.h) executes, plus the packed weight blob and
the activation-arena layout.

The host side mirrors the reference's module graph (modules/blazeface.py,
modules/mobile_facenet.py, modules/yolov5_face.py emit ops in the order the
reference's ``forward`` runs them); all arithmetic happens in the HIP kernels.
Activations are NHWC fp32 inside one arena tensor; channel counts are padded
to multiples of 4 (16-byte accesses) with zero weights in the padding.
"""
import ctypes as C
from dataclasses import dataclass

import os

import numpy as np

from . import _lib as L


def round_up(a, b):
    return (a + b - 1) // b * b


def cpad(c):
    """Internal channel count for a logical channel count (16-byte alignment)."""
    return round_up(int(c), 4)


@dataclass
class Buf:
    """An NHWC activation tensor inside the arena (offset in floats)."""
    H: int
    W: int
    C: int       # physical channels = pixel stride
    off: int
    size: int    # floats, whole batch
    ns_: int = -1   # per-image stride override (e.g. a head writing into a [B, 896, 16] tensor)
    rowpad: bool = False   # row-padded layout (include/facepath.h): off = pixel (0, 0) of image 0, row pitch (W+1)*C

    @property
    def ld(self):
        return self.C

    @property
    def ns(self):
        return self.H * self.W * self.C if self.ns_ < 0 else self.ns_

    def view(self, coff=0, C=None, cmul=1):
        return View(self, coff, self.C - coff if C is None else C, cmul)


@dataclass
class View:
    """Channel slice [coff, coff + C*cmul) of a Buf; cmul > 1 interleaves (channel_shuffle)."""
    buf: Buf
    coff: int
    C: int
    cmul: int = 1
    up: object = None   # a View of a half-size map: channels [0, up.C) of THIS view are its nearest-neighbour 2x upsampling and
                        # have NOT been written (PlanBuilder.conv folds them into the operand addressing, FP_OPF_IN_UP2, or
                        # materialises them first: PlanBuilder.materialise_up)

    @property
    def H(self):
        return self.buf.H

    @property
    def W(self):
        return self.buf.W


class Arena:
    """First-fit free-list allocator over a float arena (offsets are multiples of 64 floats)."""

    def __init__(self):
        self.free = []   # sorted list of (off, size)
        self.top = 0

    def alloc(self, size):
        size = round_up(size, 64)
        for i, (off, sz) in enumerate(self.free):
            if sz >= size:
                if sz == size:
                    self.free.pop(i)
                else:
                    self.free[i] = (off + size, sz - size)
                return off, size
        off = self.top
        self.top += size
        return off, size

    def release(self, off, size):
        self.free.append((off, size))
        self.free.sort()
        merged = []
        for o, s in self.free:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        # a free block that ends at the top lowers the top
        if merged and merged[-1][0] + merged[-1][1] == self.top:
            self.top = merged[-1][0]
            merged.pop()
        self.free = merged


def pack_conv_weight(w, cin_phys, cout_phys):
    """[Cout, Cin, KH, KW] (torch OIHW) -> Wp[Kpad/4][Npad][4], k = (ky*KW + kx)*cin_phys + ci."""
    w = np.asarray(w, dtype=np.float32)
    cout, cin, kh, kw = w.shape
    assert cin <= cin_phys and cout <= cout_phys
    K = kh * kw * cin_phys
    kpad = round_up(K, 8)
    npad = round_up(cout_phys, 32)
    full = np.zeros((kh, kw, cin_phys, npad), dtype=np.float32)
    full[:, :, :cin, :cout] = np.transpose(w, (2, 3, 1, 0))
    flat = np.zeros((kpad, npad), dtype=np.float32)
    flat[:K] = full.reshape(K, npad)
    return np.ascontiguousarray(flat.reshape(kpad // 4, 4, npad).transpose(0, 2, 1)).reshape(-1)


def pack_dw_weight(w, c_phys):
    """[C, 1, KH, KW] -> Wd[KH*KW][c_phys]."""
    w = np.asarray(w, dtype=np.float32)
    c, one, kh, kw = w.shape
    assert one == 1 and c <= c_phys
    out = np.zeros((kh * kw, c_phys), dtype=np.float32)
    out[:, :c] = w.reshape(c, kh * kw).T
    return out.reshape(-1)


def _bf16_rn_bits(x):
    """fp32 array -> uint32 bit patterns of bf16(x) << 16 (round to nearest even; no NaN / inf handling -- weights are finite)."""
    u = x.view(np.uint32)
    return (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)


def split3_bf16(w):
    """The exact three-way bf16 split of fp32 values (csrc/split.h): h = bf16(w), m = bf16(w - h), l = bf16(w - h - m), every
    conversion round-to-nearest-even, every subtraction exact, w == h + m + l.  Returns uint16 [3, ...] (the bf16 bit
    patterns of the three pieces)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    h = _bf16_rn_bits(w)
    r = w - h.view(np.float32)
    m = _bf16_rn_bits(r)
    r2 = r - m.view(np.float32)
    l = _bf16_rn_bits(r2)
    assert np.array_equal(l.view(np.float32), r2), "third piece is not exact"
    assert np.array_equal(h.view(np.float32) + m.view(np.float32) + l.view(np.float32), w)
    return np.stack([h >> 16, m >> 16, l >> 16]).astype(np.uint16)


def pad_vec(v, n, fill=0.0):
    v = np.asarray(v, dtype=np.float32).reshape(-1)
    out = np.full((n,), fill, dtype=np.float32)
    out[:v.shape[0]] = v
    return out


def bn_affine(gamma, beta, mean, var, eps):
    """Eval-mode BatchNorm as y = x*s + b (mobile_facenet.py:48-49, common.py:50)."""
    gamma, beta, mean, var = (np.asarray(t, dtype=np.float32) for t in (gamma, beta, mean, var))
    s = gamma / np.sqrt(var + np.float32(eps))
    return s.astype(np.float32), (beta - mean * s).astype(np.float32)


class PlanBuilder:
    def __init__(self, N):
        self.N = int(N)
        self.ops = []
        self.wchunks = []
        self.w_floats = 0
        self.arena = Arena()
        self.peak = 0
        self.alg_bytes = []   # op-granular algorithmic bytes per op, from LOGICAL channel counts
        self.rowpad_free = {}   # (H, W, C) -> row-padded buffers released by free()
        self.rowpad_bufs = []   # every row-padded buffer (offsets relative to the row-padded region until finish())
        self.rowpad_top = 0
        self.rowpad_end = 0
        self.has_rowpad = False
        self._placed = False

    # ---- memory ----
    def new_buf(self, H, W, C):
        C = cpad(C)
        off, size = self.arena.alloc(self.N * H * W * C)
        self.peak = max(self.peak, self.arena.top)
        return Buf(H, W, C, off, size)

    def new_buf_rowpad(self, H, W, C):
        """A buffer in the row-padded layout of include/facepath.h (one zero pixel after every row, a zero row above and
        below every image): 3x3 windows read it without bounds checks.  The pads must stay zero for the life of the
        plan, so these buffers live in a region of their own behind the recycled arena (no op ever writes there except
        through a row-padded view: a recycled block would carry another tensor's data into the pads), free() keeps
        them for the next row-padded buffer of the same shape, and the arena of such a plan starts zeroed.  Offsets are
        relative to that region until finish() places it."""
        assert not self._placed, "plan already finished"
        C = cpad(C)
        pool = self.rowpad_free.setdefault((H, W, C), [])
        if pool:
            return pool.pop()
        ns = ((H + 2) * (W + 1) + 1) * C
        base = self.rowpad_top
        self.rowpad_end = base + self.N * ns           # exact end of the region (the arena size is exact too)
        self.rowpad_top += round_up(self.N * ns, 64)
        self.has_rowpad = True
        buf = Buf(H, W, C, base + (W + 2) * C, self.N * ns, ns_=ns, rowpad=True)
        self.rowpad_bufs.append(buf)
        return buf

    def new_raw(self, floats_per_image):
        """An untyped per-image region (heads / decoded tensors); returns (offset, size) in floats."""
        off, size = self.arena.alloc(self.N * floats_per_image)
        self.peak = max(self.peak, self.arena.top)
        return off, size

    @staticmethod
    def new_shape(H, W):
        """A shape-only stand-in for an input that is not an arena buffer (u8 frames read by a *_U8 op)."""
        return Buf(H, W, 0, 0, 0)

    def free(self, buf):
        if buf.rowpad:
            self.rowpad_free.setdefault((buf.H, buf.W, buf.C), []).append(buf)
        else:
            self.arena.release(buf.off, buf.size)

    def add_weight(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        off = self.w_floats
        self.wchunks.append(arr)
        pad = round_up(arr.size, 4) - arr.size   # keep every chunk 16-byte aligned
        if pad:
            self.wchunks.append(np.zeros(pad, dtype=np.float32))
        self.w_floats += arr.size + pad
        return off

    # ---- op emission ----
    def _base(self, kind, x, out, OH, OW):
        if x.up is not None and kind != L.OP_CONV:      # only conv() knows how to read a folded upsample
            self.materialise_up(x)
        op = L.FpOp()
        op.kind = kind
        op.N = self.N
        op.H, op.W = x.H, x.W
        op.OH, op.OW = OH, OW
        op.Cin = x.C
        op.in_ld = x.buf.ld
        op.in_ns = x.buf.ns
        op.in_off = x.buf.off + x.coff
        op.out_ld = out.buf.ld
        op.out_ns = out.buf.ns
        op.out_off = out.buf.off + out.coff
        op.out_cmul = out.cmul
        op.KH = op.KW = 1
        op.stride = 1
        op.w_off = op.scale_off = op.bias_off = op.slope_off = -1
        assert x.cmul == 1, "inputs must be dense channel slices"
        assert out.H == OH and out.W == OW, (out.H, out.W, OH, OW)
        for v, bit in ((x, L.OPF_IN_ROWPAD), (out, L.OPF_OUT_ROWPAD)):
            if v.buf.rowpad:
                assert v.coff == 0 and v.C == v.buf.C and v.cmul == 1, "row-padded buffers are used whole"
                op.flags |= bit
        return op

    # Pointwise convs with K >= PW_X6_MIN_K input channels on the bf16x6 split-MFMA kernel (csrc/pwx6.hip); 0 = never.
    # (YOLOv5n-face forward at batch 256: 18.2 ms with 128, 17.0 ms with 64, 17.5 ms with 32.)
    PW_X6_MIN_K = int(os.environ.get("FP_PW_X6_MIN_K", "64"))         # (the environment variables are lab knobs)
    # Master switch of the bf16x6 split-MFMA kernels (csrc/split.h): False = every GEMM on the fp32 MFMA (the fmaf-chain
    # kernels of rounds 1-3).  Mobile-FaceNet's Depth_Wise.X6 is this attribute.
    X6 = True

    # Dense 3x3 convs (pad 1, stride 1 / 2) with at least this many input channels on the split-MFMA kernel as well
    # (csrc/pwx6.hip convx6_kernel; widths padded to 32 / 16 inside the weight planes); 0 = never.
    CONV3_X6_MIN_K = int(os.environ.get("FP_CONV3_X6_MIN_K", "32"))
    X6_SMALL_K_MIN_PIXELS = 400   # below 128 input channels only on maps of at least 20 x 20 (measured on YOLOv5-face; the
                                  # small-map 1x1 convs of BlazeFace stay on the fp32-MFMA kernels)

    # 3x3 convs on 8 / 16 / 24 input channels on the split kernel with K flattened over (tap, channel) (YOLOv5n-face's stem_2b,
    # 16 -> 32 stride 2 at 320x320: 945 us on conv_igemm_kernel)
    CONV3_X6_FLAT = os.environ.get("FP_CONV3_X6_FLAT", "1") == "1"

    @staticmethod
    def x6_tiles(cout):
        """(16-column tiles per chunk, padded width) -- mirror of general_tiles (csrc/pwx6.hip)."""
        nt = (cout + 15) // 16
        if nt <= 2:
            per = 2
        elif nt <= 3:
            per = 3
        elif nt <= 4:
            per = 4
        elif nt <= 6:
            per = 6
        else:
            p6, p4 = (nt + 5) // 6 * 6, (nt + 3) // 4 * 4
            per = 6 if p6 <= p4 else 4
        return per, (nt + per - 1) // per * per * 16

    @classmethod
    def pwx6_ok(cls, x, out, kh, kw, stride, pad, res, res_mode):
        """Mirror of fp_pwx6_eligible / fp_convx6_eligible (csrc/pwx6.hip) + the size policy above."""
        if not cls.X6 or out.cmul != 1 or x.buf.rowpad or out.buf.rowpad:
            return False
        k3 = (kh, kw) == (3, 3) and tuple(pad) == (1, 1) and stride in (1, 2) and \
            (out.H, out.W) == ((x.H + 2 - 3) // stride + 1, (x.W + 2 - 3) // stride + 1)
        k1 = (kh, kw, stride) == (1, 1, 1) and tuple(pad) == (0, 0) and (out.H, out.W) == (x.H, x.W)
        if k1:
            if not cls.PW_X6_MIN_K or x.C < cls.PW_X6_MIN_K:
                return False
        elif k3:
            flat = cls.CONV3_X6_FLAT and x.C in (8, 16, 24)      # K flattened over (tap, channel): csrc/pwx6.hip convx6_kernel
            if not cls.CONV3_X6_MIN_K or (x.C < cls.CONV3_X6_MIN_K and not flat):
                return False
        else:
            return False
        if x.C < 128 and out.H * out.W < cls.X6_SMALL_K_MIN_PIXELS:
            return False
        fast = k1 and x.C % 32 == 0 and (out.C in (48, 64) or out.C % 128 == 0) and x.buf.ns == x.H * x.W * x.buf.ld
        if x.C % 4 or out.C % 4 or out.C < (32 if k3 and x.C < 32 else 48) or (not fast and (out.H * out.W < 2 or out.W < 2)):   # (32 outputs: a third of the three-tile chunk would be padding)
            return False
        ohw = out.H * out.W
        if x.buf.ns < x.H * x.W * x.buf.ld or x.buf.ns % 4 or x.buf.ld % 4 or (x.buf.off + x.coff) % 4:
            return False
        for v in [out] + ([res] if res_mode != L.RES_NONE else []):
            if v.buf.ns != ohw * v.buf.ld or v.buf.ld % 4 or (v.buf.off + v.coff) % 4:
                return False
        if res_mode == L.RES_POOL2_BEFORE_ACT:
            return False
        if res_mode != L.RES_NONE and min(res.C, out.C) % 4:
            return False
        if res_mode == L.RES_SHUFFLE2 and (res.C < out.C or out.buf.ld < 2 * out.C):
            return False
        return True

    UP2_FOLD = os.environ.get("FP_UP2_FOLD", "1") == "1"   # nn.Upsample + Concat in front of a pointwise conv as operand addressing

    def up2_ok(self, x, out, kh, kw, stride, pad, res_mode):
        """Mirror of the FP_OPF_IN_UP2 clauses of fp_pwx6_eligible / fp_convx6_eligible (csrc/pwx6.hip): x.up can be read in
        place of the upsampled slice."""
        u = x.up
        return (self.UP2_FOLD and u is not None and res_mode == L.RES_NONE and (kh, kw, stride) == (1, 1, 1) and
                tuple(pad) == (0, 0) and u.C % 8 == 0 and 0 < u.C < x.C and u.cmul == 1 and
                x.H % 2 == 0 and x.W % 2 == 0 and (u.H, u.W) == (x.H // 2, x.W // 2) and not u.buf.rowpad and
                u.buf.ld % 4 == 0 and (u.buf.off + u.coff) % 4 == 0 and u.buf.ns % 4 == 0 and
                u.buf.ns >= u.H * u.W * u.buf.ld and self.pwx6_ok(x, out, kh, kw, stride, pad, None, res_mode))

    def materialise_up(self, x):
        """Write the upsampled slice of a view that still carries `up` (a consumer that cannot fold it)."""
        if x.up is not None:
            u, x.up = x.up, None
            self.upsample2x(u, View(x.buf, x.coff, u.C))

    @staticmethod
    def stem_dw_ok(x, w, out, stride, pad, act, res_mode):
        """Mirror of fp_stemdw_supported (csrc/stemdw.hip): Mobile-FaceNet's conv1 on a dense 112 x 112 4-float-pixel image."""
        return (tuple(w.shape) == (64, 3, 3, 3) and stride == 2 and tuple(pad) == (1, 1) and act == L.ACT_PRELU and
                res_mode == L.RES_NONE and (x.H, x.W, x.C) == (112, 112, 4) and x.buf.ld == 4 and x.coff == 0 and
                x.up is None and not x.buf.rowpad and (out.H, out.W, out.C) == (56, 56, 64) and out.cmul == 1 and
                out.coff == 0 and out.buf.ld == 64 and not out.buf.rowpad)

    def conv(self, x, w, out, stride=1, pad=(0, 0), scale=None, bias=None, slope=None,
             act=L.ACT_NONE, res=None, res_mode=L.RES_NONE, n_convs=1, out_dw=None):
        """Dense conv (OIHW weight); out is a View whose C >= Cout (extra channels get zeros).
        out_dw = (weights [C,1,3,3], (scale, bias), PReLU slope) of a depthwise 3x3 stride-1 pad-1 Conv_block computed behind
        the conv in the same kernel (FP_OPF_OUT_DW: Mobile-FaceNet's conv1 + conv2_dw); `out` then receives ITS output."""
        cout, cin, kh, kw = w.shape
        assert cin <= x.C, (cin, x.C)
        OH, OW = out.H, out.W
        fold = x.up is not None and self.up2_ok(x, out, kh, kw, stride, pad, res_mode)
        if x.up is not None and not fold:
            self.materialise_up(x)
        op = self._base(L.OP_CONV, x, out, OH, OW)
        op.Cout = out.C
        op.KH, op.KW, op.stride = kh, kw, stride
        op.pad_t, op.pad_l = pad
        op.act, op.res_mode = act, res_mode
        if fold:
            u = x.up
            op.flags |= L.OPF_IN_UP2
            op.res_ld, op.res_ns, op.res_off = u.buf.ld, u.buf.ns, u.buf.off + u.coff
            op.res_C, op.res_H, op.res_W = u.C, u.H, u.W
        if out_dw is not None and self.X6:
            # FP_OPF_OUT_DW + FP_OPF_SPLIT3 (csrc/stemdw.hip): K = (tap, channel) flattened into one 32-k slab, three bf16 planes
            # [channel tile of 16][plane][channel][32 k]
            flat = np.zeros((cout, 32), np.float32)
            flat[:, :kh * kw * cin] = np.asarray(w, np.float32).transpose(0, 2, 3, 1).reshape(cout, -1)     # k = (ky*3 + kx)*3 + c
            w3 = split3_bf16(flat).reshape(3, cout // 16, 16, 32).transpose(1, 0, 2, 3)
            op.w_off = self.add_weight(np.ascontiguousarray(w3).reshape(-1).view(np.float32))
            op.flags |= L.OPF_SPLIT3
        elif self.pwx6_ok(x, out, kh, kw, stride, pad, res, res_mode) and not (op.flags & ~L.OPF_IN_UP2):
            # three bf16 planes [tap * CS + cs][3][Npad][32] (include/facepath.h, FP_OPF_SPLIT3 on FP_OP_CONV): K runs
            # over (tap, 32-channel slab), zero rows / columns in the padding of Cin to 32 and Cout to whole chunks
            cs = (x.C + 31) // 32
            npad = self.x6_tiles(out.C)[1]
            if x.C < 32:      # flat: k = tap * Cin_phys + channel, slabs of 32 consecutive k, zero rows behind the last tap
                nsl = (kh * kw * x.C + 31) // 32
                flat = np.zeros((npad, nsl * 32), np.float32)
                wt = np.zeros((cout, kh * kw, x.C), np.float32)
                wt[:, :, :cin] = np.asarray(w, np.float32).reshape(cout, cin, kh * kw).transpose(0, 2, 1)
                flat[:cout, :kh * kw * x.C] = wt.reshape(cout, -1)
                w3 = split3_bf16(flat).reshape(3, npad, nsl, 32).transpose(2, 0, 1, 3)       # [slab][plane][n][32]
            else:
                full = np.zeros((kh * kw, npad, cs * 32), np.float32)
                full[:, :cout, :cin] = np.asarray(w, np.float32).reshape(cout, cin, kh * kw).transpose(2, 0, 1)
                w3 = split3_bf16(full).reshape(3, kh * kw, npad, cs, 32).transpose(1, 3, 0, 2, 4)
            op.w_off = self.add_weight(np.ascontiguousarray(w3).reshape(-1).view(np.float32))
            op.flags |= L.OPF_SPLIT3
        else:
            op.w_off = self.add_weight(pack_conv_weight(w, x.C, out.C))
        if cin == 3 and x.C == 4 and x.buf.ld == 4:   # 3-channel image padded to 16-byte pixels: the pad channel's weights are zero
            op.flags |= L.OPF_IN_C3
        if scale is not None:
            op.scale_off = self.add_weight(pad_vec(scale, out.C, 0.0))
        if bias is not None:
            op.bias_off = self.add_weight(pad_vec(bias, out.C, 0.0))
        if out_dw is not None:
            # the conv's slopes followed by the depthwise block [12][Cout]: nine taps, BN scale, BN bias, PReLU slope
            assert self.stem_dw_ok(x, w, out, stride, pad, act, res_mode) and slope is not None
            dw_w, dw_aff, dw_slope = out_dw
            assert tuple(dw_w.shape) == (out.C, 1, 3, 3)
            op.flags |= L.OPF_OUT_DW
            op.slope_off = self.add_weight(np.concatenate([pad_vec(slope, out.C, 0.0), pack_dw_weight(dw_w, out.C),
                                                           pad_vec(dw_aff[0], out.C), pad_vec(dw_aff[1], out.C),
                                                           pad_vec(dw_slope, out.C)]))
        elif slope is not None:
            op.slope_off = self.add_weight(pad_vec(slope, out.C, 0.0))
        if res_mode == L.RES_SHUFFLE2:   # out is the dense view of the conv's own Cout channels; 2*Cout are written
            assert out.cmul == 1 and out.coff + 2 * out.C <= out.buf.ld and res is not None and res.C >= out.C
        if res_mode != L.RES_NONE:
            assert res is not None and res.cmul == 1
            op.res_ld = res.buf.ld
            op.res_ns = res.buf.ns
            op.res_off = res.buf.off + res.coff
            op.res_C = min(res.C, out.C)
            op.res_H, op.res_W = res.H, res.W
        self.ops.append(op)
        # n_convs > 1: several reference convs on the same input merged into one op (their outputs concatenated): the
        # op-granular model counts the input once per conv
        # (FP_OPF_OUT_DW: + the depthwise conv's input and output, SURVEY 8d counts every conv)
        self.alg_bytes.append(4 * self.N * (n_convs * x.H * x.W * cin + OH * OW * cout + (2 * OH * OW * cout if out_dw is not None else 0)))
        return out

    def dwconv(self, x, w, out, stride=1, pad=(0, 0), scale=None, bias=None, slope=None, act=L.ACT_NONE):
        c, _, kh, kw = w.shape
        assert c <= x.C and out.C == x.C and out.cmul == 1
        op = self._base(L.OP_DWCONV, x, out, out.H, out.W)
        op.Cout = x.C
        op.KH, op.KW, op.stride = kh, kw, stride
        op.pad_t, op.pad_l = pad
        op.act = act
        op.w_off = self.add_weight(pack_dw_weight(w, x.C))
        if scale is not None:
            op.scale_off = self.add_weight(pad_vec(scale, x.C, 0.0))
        if bias is not None:
            op.bias_off = self.add_weight(pad_vec(bias, x.C, 0.0))
        if slope is not None:
            op.slope_off = self.add_weight(pad_vec(slope, x.C, 0.0))
        self.ops.append(op)
        self.alg_bytes.append(4 * self.N * (x.H * x.W * c + out.H * out.W * c))
        return out

    @staticmethod
    def blazeblock_lds_bytes(cin_phys, cout_phys):
        """Mirror of fp_blazeblock_lds_bytes (csrc/blaze.hip): the fused kernel needs <= 64 KiB of LDS."""
        kpad, npad = round_up(cin_phys, 8), round_up(cout_phys, 32)
        a = 128 * max(kpad + 4, cout_phys)
        return 4 * (a + 128 * (cin_phys + 4) + kpad * npad + 10 * cin_phys)

    def blazeblock(self, x, wd, bd, wp, bp, out, stride):
        """Fused BlazeBlock (blazeface.py:12-47): dw3x3(stride) -> 1x1 -> + shortcut -> ReLU in one kernel."""
        cin = wd.shape[0]
        cout = wp.shape[0]
        assert out.cmul == 1 and out.coff == 0 and out.buf.ld == out.C
        op = self._base(L.OP_BLAZEBLOCK, x, out, out.H, out.W)
        op.Cout = out.C
        op.KH = op.KW = 3
        op.stride = stride
        op.pad_t = op.pad_l = 1 if stride == 1 else 0
        op.res_C = min(cin, x.C)
        op.w_off = self.add_weight(pack_dw_weight(wd, x.C))
        op.scale_off = self.add_weight(pad_vec(bd, x.C, 0.0))
        op.slope_off = self.add_weight(pack_conv_weight(wp, x.C, out.C))
        op.bias_off = self.add_weight(pad_vec(bp, out.C, 0.0))
        self.ops.append(op)
        opix = out.H * out.W
        self.alg_bytes.append(4 * self.N * (x.H * x.W * cin + opix * cin + opix * cin + opix * cout))
        return out

    @staticmethod
    def blazepair_supported(x):
        """Mirror of fp_blazepair_supported (csrc/blazepair.hip): a row-padded 24-channel map, 128 or 64 pixels wide."""
        return (x.buf.rowpad and x.coff == 0 and x.C == 24 and x.buf.ld == 24 and x.W in (128, 64) and x.H % 8 == 0 and
                x.H >= 64)

    def blazepair(self, x, blocks, out):
        """Two consecutive stride-1 24 -> 24 BlazeBlocks (blazeface.py:12-47) as ONE op (FP_OP_BLAZEPAIR): blocks =
        ((dw_w, dw_b, pw_w, pw_b), (dw_w, dw_b, pw_w, pw_b)); the tensor between them never reaches HBM."""
        assert self.blazepair_supported(x) and len(blocks) == 2 and out.C == 24 and out.coff == 0 and out.buf.ld == 24
        op = self._base(L.OP_BLAZEPAIR, x, out, out.H, out.W)
        op.Cout = 24
        op.KH = op.KW = 3
        op.stride = 1
        op.pad_t = op.pad_l = 1
        op.act, op.res_mode, op.res_C = L.ACT_RELU, L.RES_ADD_BEFORE_ACT, 24
        op.res_ld, op.res_ns, op.res_off, op.res_H, op.res_W = op.in_ld, op.in_ns, op.in_off, x.H, x.W
        for wd, bd, wp, bp in blocks:
            assert wd.shape == (24, 1, 3, 3) and wp.shape[:2] == (24, 24)
        op.w_off = self.add_weight(np.concatenate([pack_dw_weight(b[0], 24) for b in blocks]))
        op.scale_off = self.add_weight(np.concatenate([pad_vec(b[1], 24) for b in blocks]))
        op.slope_off = self.add_weight(np.concatenate([pack_conv_weight(b[2], 24, 24) for b in blocks]))
        op.bias_off = self.add_weight(np.concatenate([pad_vec(b[3], 24) for b in blocks]))
        self.ops.append(op)
        pix = out.H * out.W
        self.alg_bytes.append(2 * 4 * self.N * pix * 24 * 4)      # two blocks, four tensor passes each
        return out

    @staticmethod
    def blazepair_s2_supported(x, cout2):
        """Mirror of fp_blazepair_s2_supported (csrc/blazepairs2.hip): a row-padded 24-channel map, 128 or 64 pixels wide, in
        front of a stride-1 24 -> 24 block followed by a stride-2 24 -> 24 / 48 block."""
        return (x.buf.rowpad and x.coff == 0 and x.C == 24 and x.buf.ld == 24 and x.W in (128, 64) and x.H % 8 == 0 and
                x.H >= 16 and cout2 in (24, 48))      # H / 2 output rows in bands of a multiple of 4, at least two bands

    def blazepair_s2(self, x, blocks, out):
        """A stride-1 24 -> 24 BlazeBlock and the STRIDE-2 BlazeBlock behind it (blazeface.py:12-47) as ONE op (FP_OP_BLAZEPAIR
        with stride = 2): blocks = ((dw_w, dw_b, pw_w, pw_b) of the stride-1 block, the same of the stride-2 block); the
        full-size tensor between them never reaches HBM, `out` is the half-size map (dense or row-padded, ld = its channels)."""
        cout2 = blocks[1][2].shape[0]
        assert self.blazepair_s2_supported(x, cout2) and len(blocks) == 2
        assert out.C == cout2 and out.coff == 0 and out.buf.ld == cout2 and (out.H, out.W) == (x.H // 2, x.W // 2)
        op = self._base(L.OP_BLAZEPAIR, x, out, out.H, out.W)
        op.Cout = cout2
        op.KH = op.KW = 3
        op.stride = 2
        op.pad_t = op.pad_l = 0
        op.act, op.res_mode, op.res_C = L.ACT_RELU, L.RES_POOL2_BEFORE_ACT, 24
        op.res_ld, op.res_ns, op.res_off, op.res_H, op.res_W = op.in_ld, op.in_ns, op.in_off, x.H, x.W
        (wd1, bd1, wp1, bp1), (wd2, bd2, wp2, bp2) = blocks
        assert wd1.shape == (24, 1, 3, 3) and wp1.shape[:2] == (24, 24) and wd2.shape == (24, 1, 3, 3) and wp2.shape[:2] == (cout2, 24)
        op.w_off = self.add_weight(np.concatenate([pack_dw_weight(wd1, 24), pack_dw_weight(wd2, 24)]))
        op.scale_off = self.add_weight(np.concatenate([pad_vec(bd1, 24), pad_vec(bd2, 24)]))
        op.slope_off = self.add_weight(np.concatenate([pack_conv_weight(wp1, 24, 24), pack_conv_weight(wp2, 24, cout2)]))
        op.bias_off = self.add_weight(np.concatenate([pad_vec(bp1, 24), pad_vec(bp2, cout2)]))
        self.ops.append(op)
        pix, opix = x.H * x.W, out.H * out.W
        # the stride-1 block's four tensor passes + the stride-2 block's (input, dw output, 1x1 input, output)
        self.alg_bytes.append(4 * self.N * (pix * 24 * 4 + pix * 24 + opix * 24 + opix * 24 + opix * cout2))
        return out

    @staticmethod
    def blazechain_supported(x):
        """Mirror of fp_blazechain_supported (csrc/blazechain.hip): a dense 96-channel 16 x 16 map."""
        return (not x.buf.rowpad and x.coff == 0 and x.C == 96 and x.buf.ld == 96 and x.cmul == 1 and x.H == 16 and x.W == 16)

    def blazechain(self, x, blocks, out):
        """A run of stride-1 96 -> 96 BlazeBlocks on the 16 x 16 map (blazeface.py:12-47,146-152) as ONE op
        (FP_OP_BLAZECHAIN): blocks = ((dw_w, dw_b, pw_w, pw_b), ...); the tensors between them never reach HBM.  The 1x1
        weights go in as three bf16 planes (split3_bf16), one slab per 32 input channels: include/facepath.h BLAZECHAIN."""
        assert self.blazechain_supported(x) and 1 <= len(blocks) <= 16
        assert out.C == 96 and out.coff == 0 and out.buf.ld == 96 and not out.buf.rowpad and (out.H, out.W) == (16, 16)
        op = self._base(L.OP_BLAZECHAIN, x, out, 16, 16)
        op.Cout = 96
        op.KH = op.KW = 3
        op.stride = 1
        op.pad_t = op.pad_l = 1
        op.act, op.res_mode, op.res_C = L.ACT_RELU, L.RES_ADD_BEFORE_ACT, 96
        op.res_ld, op.res_ns, op.res_off, op.res_H, op.res_W = op.in_ld, op.in_ns, op.in_off, 16, 16
        op.Cmid = len(blocks)
        op.flags |= L.OPF_SPLIT3
        chunks = []
        for wd, bd, wp, bp in blocks:
            assert wd.shape == (96, 1, 3, 3) and wp.shape[:2] == (96, 96)
            par = np.zeros(1280, np.float32)
            par[:864] = pack_dw_weight(wd, 96)
            par[864:960] = pad_vec(bd, 96)
            par[960:1056] = pad_vec(bp, 96)
            w3 = split3_bf16(np.asarray(wp, np.float32).reshape(96, 96))          # [3][cout][cin]
            w3 = w3.reshape(3, 96, 3, 32).transpose(2, 0, 1, 3)                    # [slab][plane][cout][32]
            chunks += [par, np.ascontiguousarray(w3).reshape(-1).view(np.float32)]
        op.w_off = self.add_weight(np.concatenate(chunks))
        self.ops.append(op)
        self.alg_bytes.append(len(blocks) * 4 * self.N * 256 * 96 * 4)     # four tensor passes per block
        return out

    DWPW_X6 = os.environ.get("FP_DWPW_X6", "1") == "1"    # the dw -> 1x1 op with its 1x1 on the split MFMA (csrc/dwpwx6.hip)

    @classmethod
    def dwpwx6_ok(cls, x, out, G, stride, res, shuffle, out_slope, out_act):
        """Mirror of fp_dwpwx6_eligible (csrc/dwpwx6.hip)."""
        if not (cls.X6 and cls.DWPW_X6) or out_slope is not None or out_act not in (L.ACT_NONE, L.ACT_SILU) or stride not in (1, 2):
            return False
        if G % 32 or G > 256 or out.C not in (64, 128) or out.W % 4 or out.cmul != 1 or x.buf.rowpad or out.buf.rowpad:
            return False
        if out.C != 128:      # measured on YOLOv5n-face (256 images): 128 -> 128 at 40x40 278 -> 246 us, 80x80 stride 2 367 / 298 ->
            return False      # 345 / 248; 64 -> 64 at 80x80 420 -> 440 (the depthwise phase, not the 1x1, bounds the narrow form)
        ohw = out.H * out.W
        if x.buf.ld % 4 or (x.buf.off + x.coff) % 4 or x.buf.ns % 4 or x.buf.ns < x.H * x.W * x.buf.ld:
            return False
        for v in [out] + ([res] if res is not None else []):
            if v.buf.ld % 4 or (v.buf.off + v.coff) % 4 or v.buf.ns != ohw * v.buf.ld:
                return False
        if res is not None and min(res.C, out.C) % 4:
            return False
        if shuffle and (res.C < out.C or out.buf.ld < 2 * out.C):
            return False
        return True

    def dwpw(self, x, dw_w, dw_scale, dw_bias, dw_slope, pw_w, pw_scale, pw_bias, out, stride, res=None,
             out_slope=None, out_act=L.ACT_NONE, shuffle=False):
        """Fused Depth_Wise tail (mobile_facenet.py:72-85): dw3x3 stride s (+BN affine, +PReLU) -> 1x1 (+BN affine)
        [+ res], or -- with out_slope -- a depthwise Conv_block followed by a 1x1 Conv_block (BN + PReLU on both:
        conv2_dw -> conv_23.conv, mobile_facenet.py:117-118,70).  x has G (multiple of 64) channels."""
        G = dw_w.shape[0]
        cout, cin = pw_w.shape[0], pw_w.shape[1]
        assert cin == G == x.C and G % 64 == 0 and out.cmul == 1
        # out_act = ACT_SILU: SiLU on the 1x1 output; shuffle: `out` is the dense view of the conv's own Cout channels
        # inside a buffer that receives 2*Cout (out[2n] = res[n], out[2n+1] = y[n]: ShuffleV2Block's cat + shuffle)
        if shuffle:
            assert res is not None and res.C >= out.C and out.coff + 2 * out.C <= out.buf.ld
        else:
            assert out.coff == 0 and out.buf.ld == out.C or out_act != L.ACT_NONE
        op = self._base(L.OP_DWPW, x, out, out.H, out.W)
        split = self.dwpwx6_ok(x, out, G, stride, res, shuffle, out_slope, out_act)
        op.act2 = out_act
        op.Cout = out.C
        op.KH = op.KW = 3
        op.stride = stride
        op.pad_t = op.pad_l = 1
        op.act = L.ACT_PRELU if dw_slope is not None else L.ACT_NONE
        slope = dw_slope if dw_slope is not None else np.zeros(G, np.float32)
        op.w_off = self.add_weight(np.concatenate([pack_dw_weight(dw_w, G), pad_vec(dw_scale, G), pad_vec(dw_bias, G),
                                                   pad_vec(slope, G)]))
        c4 = round_up(out.C, 4)
        if split:
            # the 1x1 as three bf16 planes [G / 32][3][N][32] (csrc/dwpwx6.hip), then [N] BN scale, [N] BN bias
            full = np.zeros((out.C, G), np.float32)
            full[:cout] = np.asarray(pw_w, np.float32).reshape(cout, G)
            w3 = split3_bf16(full).reshape(3, out.C, G // 32, 32).transpose(2, 0, 1, 3)
            op.flags |= L.OPF_SPLIT3
            op.slope_off = self.add_weight(np.concatenate([np.ascontiguousarray(w3).reshape(-1).view(np.float32),
                                                           pad_vec(pw_scale, c4), pad_vec(pw_bias, c4)]))
        else:
            op.slope_off = self.add_weight(np.concatenate([pack_conv_weight(pw_w, G, out.C), pad_vec(pw_scale, c4),
                                                           pad_vec(pw_bias, c4)]))
        if out_slope is not None:
            assert res is None
            op.bias_off = self.add_weight(pad_vec(out_slope, c4))
        if res is not None:
            assert res.cmul == 1
            op.res_mode = L.RES_SHUFFLE2 if shuffle else L.RES_ADD_AFTER_ACT
            op.res_ld, op.res_ns = res.buf.ld, res.buf.ns
            op.res_off = res.buf.off + res.coff
            op.res_C = min(res.C, out.C)
            op.res_H, op.res_W = res.H, res.W
        self.ops.append(op)
        opix = out.H * out.W
        self.alg_bytes.append(4 * self.N * (x.H * x.W * G + opix * G + opix * G + opix * cout))
        return out

    # shapes csrc/dwblock.hip is instantiated for: (block width, map size); Cmid = 2 * width
    DWBLOCK_SHAPES = ((128, 14), (128, 7), (64, 28))

    @classmethod
    def dwblock_supported(cls, x, cin, cmid, cout, stride):
        """Mirror of fp_dwblock_supported (csrc/dwblock.hip): stride-1 Depth_Wise blocks on a dense square map."""
        return (stride == 1 and cin == cout and cmid == 2 * cin and x.H == x.W and (cin, x.H) in cls.DWBLOCK_SHAPES and
                x.coff == 0 and x.C == cin and x.buf.ld == cin and x.buf.ns == x.H * x.W * cin and not x.buf.rowpad)

    # shapes csrc/dwblockx6.hip (bf16x6 split MFMA, OPF_SPLIT3) is instantiated for
    DWBLOCK_X6_SHAPES = ((128, 14), (128, 7), (64, 28))
    # ... and its stride-2 form (dwblock_x6d_kernel): (Cin, Cmid, Cout, input map size)
    DWBLOCK_X6D_SHAPES = ((64, 256, 128, 28), (128, 512, 128, 14), (64, 128, 64, 56))

    @classmethod
    def dwblock_x6d_supported(cls, x, cin, cmid, cout, stride):
        return (stride == 2 and x.H == x.W and (cin, cmid, cout, x.H) in cls.DWBLOCK_X6D_SHAPES and x.coff == 0 and x.C == cin and
                x.buf.ld == cin and x.buf.ns == x.H * x.W * cin and not x.buf.rowpad)

    def dwblock(self, x, e_w, e_aff, e_slope, dw_w, dw_aff, dw_slope, pw_w, pw_aff, out, residual, split=False, stride=1,
                in_dw=None):
        """A whole Depth_Wise block (mobile_facenet.py:67-88) as ONE op (FP_OP_DWBLOCK, csrc/dwblock.hip): 1x1 expand
        + BN + PReLU -> dw3x3 (stride 1) + BN + PReLU -> 1x1 project + BN [+ x]; the expanded tensor stays in LDS.
        *_aff = (scale, bias) of the eval-mode BatchNorm."""
        cmid, cin = e_w.shape[0], e_w.shape[1]
        cout = pw_w.shape[0]
        if stride == 1:
            assert self.dwblock_supported(x, cin, cmid, cout, 1)
        else:   # stride-2 blocks exist only in the split-MFMA form
            assert split and not residual and self.dwblock_x6d_supported(x, cin, cmid, cout, stride)
        assert dw_w.shape == (cmid, 1, 3, 3) and pw_w.shape[1] == cmid and (out.H, out.W) == (x.H // stride, x.W // stride)
        assert out.coff == 0 and out.C == cout and out.buf.ld == cout and out.cmul == 1 and not out.buf.rowpad
        op = self._base(L.OP_DWBLOCK, x, out, out.H, out.W)
        op.Cout, op.Cmid = cout, cmid
        op.KH = op.KW = 3
        op.stride = stride
        op.pad_t = op.pad_l = 1
        op.act = L.ACT_PRELU
        if split:
            # three bf16 planes per matrix, in the fragment order of dwblock_x6_kernel (include/facepath.h, DWBLOCK)
            assert stride == 2 or (cin, x.H) in self.DWBLOCK_X6_SHAPES
            op.flags |= L.OPF_SPLIT3
            R = cmid // 32
            e3 = split3_bf16(np.asarray(e_w, np.float32).reshape(cmid, cin))            # [3][g][k]
            e3 = e3.reshape(3, R, 32, cin // 32, 32).transpose(1, 0, 3, 2, 4)            # [R][3][ks][g'][k']
            p3 = split3_bf16(np.asarray(pw_w, np.float32).reshape(cout, cmid))           # [3][co][g]
            p3 = p3.reshape(3, cout, R, 32).transpose(2, 0, 1, 3)                        # [R][3][co][g']
            op.w_off = self.add_weight(np.ascontiguousarray(e3).reshape(-1).view(np.float32))
            wp = np.ascontiguousarray(p3).reshape(-1).view(np.float32)
        else:
            op.w_off = self.add_weight(pack_conv_weight(e_w, cin, cmid))
            wp = pack_conv_weight(pw_w, cmid, cout)
        rows = [pad_vec(e_aff[0], cmid), pad_vec(e_aff[1], cmid), pad_vec(e_slope, cmid), pack_dw_weight(dw_w, cmid),
                pad_vec(dw_aff[0], cmid), pad_vec(dw_aff[1], cmid), pad_vec(dw_slope, cmid)]
        op.scale_off = self.add_weight(np.concatenate(rows))
        op.slope_off = self.add_weight(np.concatenate([wp, pad_vec(pw_aff[0], cout), pad_vec(pw_aff[1], cout)]))
        if residual:
            op.res_mode = L.RES_ADD_AFTER_ACT
            op.res_ld, op.res_ns, op.res_off = op.in_ld, op.in_ns, op.in_off
            op.res_C, op.res_H, op.res_W = cin, x.H, x.W
        self.ops.append(op)
        pix, opix = x.H * x.W, out.H * out.W
        extra = 0
        if in_dw is not None:
            # OPF_IN_DW: a depthwise 3x3 stride-1 Conv_block (weights [C,1,3,3], (scale, bias), PReLU slope) in front of the
            # block, computed in the kernel's prologue (conv2_dw + conv_23 of Mobile-FaceNet); parameters [12][Cin] at bias_off
            iw, iaff, islope = in_dw
            assert split and stride == 2 and (cin, cmid, cout, x.H) == (64, 128, 64, 56) and iw.shape == (cin, 1, 3, 3)
            op.flags |= L.OPF_IN_DW
            op.bias_off = self.add_weight(np.concatenate([pack_dw_weight(iw, cin), pad_vec(iaff[0], cin), pad_vec(iaff[1], cin),
                                                          pad_vec(islope, cin)]))
            extra = pix * 2 * cin
        # the three convs of the block (+ the depthwise conv in front), each input once + output once
        self.alg_bytes.append(4 * self.N * (extra + pix * (cin + cmid) + (pix + opix) * cmid + opix * (cmid + cout)))
        return out

    # shapes csrc/shufdown.hip is instantiated for: (Cin, branch width)
    SHUFDOWN_SHAPES = ((32, 64),)

    @classmethod
    def shufdown_supported(cls, x, out, cin, cb):
        """Mirror of fp_shufdown_supported (csrc/shufdown.hip): a whole stride-2 ShuffleV2Block as one op."""
        if not cls.X6 or (cin, cb) not in cls.SHUFDOWN_SHAPES or x.C != cin or out.C != 2 * cb or out.cmul != 1:
            return False
        if x.H % 2 or x.W % 2 or (out.H, out.W) != (x.H // 2, x.W // 2) or x.buf.rowpad or out.buf.rowpad:
            return False
        for v in (x, out):
            if v.buf.ld % 4 or (v.buf.off + v.coff) % 4 or v.buf.ns % 4 or v.buf.ns < v.H * v.W * v.buf.ld:
                return False
        return True

    def shufdown(self, x, b1_dw, b1_dw_aff, b1_pw, b1_pw_aff, pw1, pw1_aff, dw2, dw2_aff, pw2, pw2_aff, out):
        """A whole stride-2 ShuffleV2Block (y5/models/common.py:127-176) as ONE op (FP_OP_SHUFDOWN, csrc/shufdown.hip):
        branch1 = dw3x3 s2 + BN -> 1x1 + BN + SiLU, branch2 = 1x1 + BN + SiLU -> dw3x3 s2 + BN -> 1x1 + BN + SiLU,
        out[2c] = branch1[c], out[2c + 1] = branch2[c].  *_aff = (scale, bias) of the eval-mode BatchNorm.  The parameter block's
        layout is facepath.h "SHUFDOWN"."""
        cin, cb = x.C, pw1.shape[0]
        assert self.shufdown_supported(x, out, cin, cb)
        assert b1_dw.shape == (cin, 1, 3, 3) and b1_pw.shape[:2] == (cb, cin) and pw1.shape[:2] == (cb, cin)
        assert dw2.shape == (cb, 1, 3, 3) and pw2.shape[:2] == (cb, cb)
        op = self._base(L.OP_SHUFDOWN, x, out, out.H, out.W)
        op.Cout, op.Cmid = 2 * cb, cb
        op.KH = op.KW = 3
        op.stride = 2
        op.pad_t = op.pad_l = 1
        op.act = op.act2 = L.ACT_SILU
        op.flags |= L.OPF_SPLIT3
        ks, r = cin // 32, cb // 32

        def planes(a):
            return np.ascontiguousarray(a).reshape(-1).view(np.float32)
        w_b1 = split3_bf16(np.asarray(b1_pw, np.float32).reshape(cb, cin)).reshape(3, cb, ks, 32).transpose(2, 0, 1, 3)   # [ks][3][co][k']
        w_1 = split3_bf16(np.asarray(pw1, np.float32).reshape(cb, cin)).reshape(3, r, 32, ks, 32).transpose(1, 0, 3, 2, 4)  # [r][3][ks][g'][k']
        w_2 = split3_bf16(np.asarray(pw2, np.float32).reshape(cb, cb)).reshape(3, cb, r, 32).transpose(2, 0, 1, 3)           # [r][3][co][g']
        blob = [pack_dw_weight(b1_dw, cin), pad_vec(b1_dw_aff[0], cin), pad_vec(b1_dw_aff[1], cin),
                planes(w_b1), pad_vec(b1_pw_aff[0], cb), pad_vec(b1_pw_aff[1], cb),
                planes(w_1), pad_vec(pw1_aff[0], cb), pad_vec(pw1_aff[1], cb),
                pack_dw_weight(dw2, cb), pad_vec(dw2_aff[0], cb), pad_vec(dw2_aff[1], cb),
                planes(w_2), pad_vec(pw2_aff[0], cb), pad_vec(pw2_aff[1], cb)]
        op.w_off = self.add_weight(np.concatenate(blob))
        self.ops.append(op)
        pix, opix = x.H * x.W, out.H * out.W
        # the five convs of the block, each input once + output once
        self.alg_bytes.append(4 * self.N * ((pix + opix) * cin + opix * (cin + cb) + pix * (cin + cb) + (pix + opix) * cb + opix * 2 * cb))
        return out

    # branch widths csrc/shufdown.hip's stride-1 kernel (shufunit_x6_kernel) is instantiated for
    SHUFUNIT_WIDTHS = (64,)

    @classmethod
    def shufunit_supported(cls, x, out, cb):
        """Mirror of fp_shufunit_supported (csrc/shufdown.hip): a whole stride-1 ShuffleV2Block as one op."""
        if not cls.X6 or cb not in cls.SHUFUNIT_WIDTHS or x.C != 2 * cb or out.C != 2 * cb or out.cmul != 1 or x.buf is out.buf:
            return False
        if (out.H, out.W) != (x.H, x.W) or x.buf.rowpad or out.buf.rowpad:
            return False
        for v in (x, out):
            if v.buf.ld % 4 or (v.buf.off + v.coff) % 4 or v.buf.ns % 4 or v.buf.ns < v.H * v.W * v.buf.ld:
                return False
        return True

    def shufunit(self, x, pw1, pw1_aff, dw2, dw2_aff, pw2, pw2_aff, out):
        """A whole stride-1 ShuffleV2Block (y5/models/common.py:127-176) as ONE op (FP_OP_SHUFUNIT, csrc/shufdown.hip):
        x1, x2 = x.chunk(2); branch2(x2) = 1x1 + BN + SiLU -> dw3x3 + BN -> 1x1 + BN + SiLU; out[2c] = x1[c], out[2c + 1] = branch2[c].
        The parameter block's layout is facepath.h "SHUFUNIT"."""
        cb = pw1.shape[0]
        assert self.shufunit_supported(x, out, cb)
        assert pw1.shape[:2] == (cb, cb) and dw2.shape == (cb, 1, 3, 3) and pw2.shape[:2] == (cb, cb)
        op = self._base(L.OP_SHUFUNIT, x, out, out.H, out.W)
        op.Cout, op.Cmid = 2 * cb, cb
        op.KH = op.KW = 3
        op.stride = 1
        op.pad_t = op.pad_l = 1
        op.act = op.act2 = L.ACT_SILU
        op.flags |= L.OPF_SPLIT3
        ks = r = cb // 32

        def planes(a):
            return np.ascontiguousarray(a).reshape(-1).view(np.float32)
        w_1 = split3_bf16(np.asarray(pw1, np.float32).reshape(cb, cb)).reshape(3, r, 32, ks, 32).transpose(1, 0, 3, 2, 4)   # [r][3][ks][g'][k']
        w_2 = split3_bf16(np.asarray(pw2, np.float32).reshape(cb, cb)).reshape(3, cb, r, 32).transpose(2, 0, 1, 3)           # [r][3][co][g']
        blob = [planes(w_1), pad_vec(pw1_aff[0], cb), pad_vec(pw1_aff[1], cb),
                pack_dw_weight(dw2, cb), pad_vec(dw2_aff[0], cb), pad_vec(dw2_aff[1], cb),
                planes(w_2), pad_vec(pw2_aff[0], cb), pad_vec(pw2_aff[1], cb)]
        op.w_off = self.add_weight(np.concatenate(blob))
        self.ops.append(op)
        pix = x.H * x.W
        self.alg_bytes.append(4 * self.N * pix * 6 * cb)     # three convs of cb channels, input once + output once each
        return out

    @staticmethod
    def pack_stem5_x6(w):
        """[24, 3, 5, 5] -> the three bf16 planes of stem5_u8_x6_kernel (csrc/stem.hip): [3 slabs][2 channel tiles][3 planes][16][32]
        with k = 16 (ky - 2 slab) + 3 kx + c inside a slab (every ky padded to 16, the sixth ky and channels 24 .. 31 zero)."""
        w = np.asarray(w, dtype=np.float32)
        assert w.shape == (24, 3, 5, 5)
        full = np.zeros((3, 2, 16, 32), dtype=np.float32)               # [slab][nt][channel][k]
        for ky in range(5):
            for kx in range(5):
                for c in range(3):
                    k = 16 * (ky % 2) + 3 * kx + c
                    for co in range(24):
                        full[ky // 2, co // 16, co % 16, k] = w[co, c, ky, kx]
        planes = split3_bf16(full)                                      # [3 planes][slab][nt][16][32]
        blob = np.ascontiguousarray(planes.transpose(1, 2, 0, 3, 4))    # [slab][nt][plane][16][32]
        return blob.reshape(-1).view(np.float32)

    def stem_u8(self, u8, w, out, pad=(0, 0), scale=None, bias=None, slope=None, act=L.ACT_NONE, split=False):
        """First conv of a network reading u8 frames itself (FP_OP_STEM_U8): KxK (3 or 5) stride 2, Cout <= 64, dense
        output buffer.  u8 = (H, W, frame_h, frame_w, ext_index): the H x W letterbox canvas is resampled from the
        frames while the conv's input tile is staged (external buffers ext_index..+2 = frames, tap tables, LUT).
        w is the [Cout, 3, K, K] weight; it is packed for a 4-channel pixel like the fp32-canvas form."""
        H, W, fh, fw, ext_index = u8
        cout, cin, kh, kw = w.shape
        assert cin == 3 and kh == kw and kh in (3, 5) and out.cmul == 1 and out.coff == 0 and out.buf.ld == out.C
        assert out.C <= 64 and H + W <= 2048
        op = L.FpOp()
        op.kind, op.N, op.H, op.W, op.OH, op.OW = L.OP_STEM_U8, self.N, H, W, out.H, out.W
        op.Cin, op.in_ld, op.in_ns, op.in_off = 3, 3, fh * fw * 3, ext_index
        op.Cout, op.out_ld, op.out_ns, op.out_off, op.out_cmul = out.C, out.buf.ld, out.buf.ns, out.buf.off, 1
        if out.buf.rowpad:
            op.flags |= L.OPF_OUT_ROWPAD
        op.KH = op.KW = kh
        op.stride = 2
        op.pad_t, op.pad_l = pad
        op.act = act
        op.res_H, op.res_W = fh, fw
        op.w_off = op.scale_off = op.bias_off = op.slope_off = -1
        if split:
            # BlazeFace's 5x5 stem on the bf16 matrix cores (FP_OPF_SPLIT3, stem5_u8_x6_kernel): the band form's shape only
            assert (kh, H, W, out.H, out.W, cout) == (5, 256, 256, 128, 128, 24) and pad == (1, 1) and scale is None
            assert bias is not None and act == L.ACT_RELU and self.N >= 16
            op.flags |= L.OPF_SPLIT3
            op.w_off = self.add_weight(self.pack_stem5_x6(w))
        else:
            op.w_off = self.add_weight(pack_conv_weight(w, 4, out.C))
        if scale is not None:
            op.scale_off = self.add_weight(pad_vec(scale, out.C, 0.0))
        if bias is not None:
            op.bias_off = self.add_weight(pad_vec(bias, out.C, 0.0))
        if slope is not None:
            op.slope_off = self.add_weight(pad_vec(slope, out.C, 0.0))
        self.ops.append(op)
        self.alg_bytes.append(4 * self.N * (H * W * 3 + out.H * out.W * cout))
        return out

    def ystem(self, x, w1, scale1, bias1, w2, scale2, bias2, a_out, pool_out, u8=None):
        """Head of YOLOv5-face's StemBlock (common.py:58-73) as ONE op: stem_1 (3x3 s2 p1, SiLU) stays in LDS,
        stem_2a (1x1, SiLU) -> a_out, maxpool2x2(stem_1) -> pool_out (a channel slice of stem_3's concat buffer).
        scale1 / scale2 = None when the BatchNorm is folded into the conv (Model.fuse())."""
        c1, c2 = w1.shape[0], w2.shape[0]
        assert w1.shape[2:] == (3, 3) and w2.shape[1] == c1 and w2.shape[2:] == (1, 1)
        assert c1 <= 32 and a_out.C <= 32 and pool_out.C >= c1 and pool_out.cmul == 1 and a_out.cmul == 1
        if u8 is None:
            assert x.C == 4 and x.buf.ld == 4 and x.coff == 0
            H, W = x.H, x.W
            op = self._base(L.OP_YSTEM, x, a_out, H // 2, W // 2)
            if w1.shape[1] == 3:   # 3-channel image in 16-byte pixels: the pad channel's weights are zero
                op.flags |= L.OPF_IN_C3
        else:
            # u8 = (H, W, frame_h, frame_w, ext_index): the H x W canvas is never materialised; the op reads the frames
            # (external buffers ext_index .. ext_index + 2: frames, tap tables, LUT) through fp_plan_run_ext
            assert x is None
            H, W, fh, fw, ext_index = u8
            op = L.FpOp()
            op.kind, op.N, op.H, op.W, op.OH, op.OW = L.OP_YSTEM_U8, self.N, H, W, H // 2, W // 2
            op.Cin, op.in_ld, op.in_ns, op.in_off = 3, 3, fh * fw * 3, ext_index
            op.out_ld, op.out_ns, op.out_off, op.out_cmul = a_out.buf.ld, a_out.buf.ns, a_out.buf.off + a_out.coff, 1
            op.w_off = op.scale_off = op.bias_off = op.slope_off = -1
        H1, W1 = H // 2, W // 2
        assert H % 4 == 0 and W % 4 == 0 and (a_out.H, a_out.W) == (H1, W1) and (pool_out.H, pool_out.W) == (H1 // 2, W1 // 2)
        op.Cout = a_out.C
        op.KH = op.KW = 3
        op.stride = 2
        op.pad_t = op.pad_l = 1
        op.act = L.ACT_SILU
        op.res_ld, op.res_ns = pool_out.buf.ld, pool_out.buf.ns
        op.res_off = pool_out.buf.off + pool_out.coff
        op.res_C = cpad(c1)
        op.res_H, op.res_W = (pool_out.H, pool_out.W) if u8 is None else (u8[2], u8[3])
        op.w_off = self.add_weight(pack_conv_weight(w1, 4, cpad(c1)))
        if scale1 is not None:
            op.scale_off = self.add_weight(pad_vec(scale1, 32, 0.0))
        op.bias_off = self.add_weight(pad_vec(bias1, 32, 0.0))
        nb2 = (a_out.C + 15) // 16
        wq = np.zeros((32, nb2 * 16), np.float32)                       # [k][n], zero padded
        wq[:c1, :c2] = np.asarray(w2, np.float32).reshape(c2, c1).T
        blob = [np.ascontiguousarray(wq.reshape(2, 4, 4, nb2 * 16).transpose(0, 1, 3, 2)).reshape(-1),  # [j][g][n][e]
                pad_vec(scale2 if scale2 is not None else np.ones(c2, np.float32), nb2 * 16, 0.0),
                pad_vec(bias2, nb2 * 16, 0.0)]
        op.slope_off = self.add_weight(np.concatenate(blob))
        self.ops.append(op)
        self.alg_bytes.append(4 * self.N * (H * W * 3 + H1 * W1 * c1 + H1 * W1 * c1 + H1 * W1 * c2))
        return a_out

    @classmethod
    def ystem2_supported(cls, a, pool, out):
        """Mirror of fp_ystem2_supported (csrc/ystem2.hip): stem_2b + cat + stem_3 of YOLOv5n-face's StemBlock (c = 32) as one op."""
        if not cls.X6 or a.C != 16 or pool.C != 32 or out.C != 32 or out.cmul != 1 or a.H % 2 or a.W % 2:
            return False
        if (out.H, out.W) != (a.H // 2, a.W // 2) or (pool.H, pool.W) != (out.H, out.W):
            return False
        for v in (a, pool, out):
            if v.buf.rowpad or v.buf.ld % 4 or (v.buf.off + v.coff) % 4 or v.buf.ns % 4 or v.buf.ns < v.H * v.W * v.buf.ld:
                return False
        return out.buf is not a.buf and out.buf is not pool.buf

    def ystem2(self, a, pool, w2b, aff2b, w3, aff3, out):
        """The tail of YOLOv5-face's StemBlock (y5/models/common.py:58-73) as ONE op (FP_OP_YSTEM2, csrc/ystem2.hip):
        out = stem_3(cat(stem_2b(a), pool)), both convs + (BN) + SiLU.  *_aff = (scale or None, bias).  Layout: facepath.h "YSTEM2"."""
        assert self.ystem2_supported(a, pool, out) and w2b.shape == (32, 16, 3, 3) and w3.shape[:2] == (32, 64)
        op = self._base(L.OP_YSTEM2, a, out, out.H, out.W)
        op.Cout = 32
        op.KH = op.KW = 3
        op.stride = 2
        op.pad_t = op.pad_l = 1
        op.act = op.act2 = L.ACT_SILU
        op.flags |= L.OPF_SPLIT3
        op.res_ld, op.res_ns = pool.buf.ld, pool.buf.ns
        op.res_off = pool.buf.off + pool.coff
        op.res_C, op.res_H, op.res_W = 32, pool.H, pool.W

        def planes(x):
            return np.ascontiguousarray(x).reshape(-1).view(np.float32)

        def aff(sb):
            return [np.ones(32, np.float32) if sb[0] is None else pad_vec(sb[0], 32), pad_vec(sb[1], 32)]
        k2 = np.zeros((32, 160), np.float32)                                    # k = (ky*3 + kx)*16 + c, padded to five slabs
        k2[:, :144] = np.asarray(w2b, np.float32).transpose(0, 2, 3, 1).reshape(32, 144)
        p2 = split3_bf16(k2).reshape(3, 32, 5, 32).transpose(2, 0, 1, 3)        # [slab][3][co][k']
        p3 = split3_bf16(np.asarray(w3, np.float32).reshape(32, 64)).reshape(3, 32, 2, 32).transpose(2, 0, 1, 3)
        op.w_off = self.add_weight(np.concatenate([planes(p2)] + aff(aff2b) + [planes(p3)] + aff(aff3)))
        self.ops.append(op)
        pix, opix = a.H * a.W, out.H * out.W
        self.alg_bytes.append(4 * self.N * (pix * 16 + opix * 32 + opix * 64 + opix * 32))   # the two convs
        return out

    def maxpool(self, x, out, k, stride, pad):
        assert out.C == x.C
        op = self._base(L.OP_MAXPOOL, x, out, out.H, out.W)
        op.Cout = x.C
        op.KH = op.KW = k
        op.stride = stride
        op.pad_t = op.pad_l = pad
        self.ops.append(op)
        self.alg_bytes.append(0)
        return out

    def upsample2x(self, x, out):
        assert out.C == x.C and out.H == 2 * x.H and out.W == 2 * x.W
        op = self._base(L.OP_UPSAMPLE2X, x, out, out.H, out.W)
        op.Cout = x.C
        self.ops.append(op)
        self.alg_bytes.append(0)
        return out

    def copy(self, x, out):
        assert out.C == x.C and out.H == x.H and out.W == x.W
        op = self._base(L.OP_COPY, x, out, out.H, out.W)
        op.Cout = x.C
        self.ops.append(op)
        self.alg_bytes.append(0)
        return out

    def l2norm(self, x, out):
        op = self._base(L.OP_L2NORM, x, out, out.H, out.W)
        op.Cout = x.C
        self.ops.append(op)
        self.alg_bytes.append(0)
        return out

    def finish(self):
        if not self._placed:   # the row-padded region goes behind the recycled arena: relocate its views once
            self._placed = True
            if self.rowpad_top:
                shift = round_up(self.peak, 64)
                for op in self.ops:
                    if op.flags & L.OPF_IN_ROWPAD:
                        if op.kind == L.OP_BLAZEPAIR:
                            op.res_off += shift      # its shortcut view IS its input view
                        op.in_off += shift
                    if op.flags & L.OPF_OUT_ROWPAD:
                        op.out_off += shift
                for buf in self.rowpad_bufs:
                    buf.off += shift
                self.peak = shift + self.rowpad_end
        weights = np.concatenate(self.wchunks) if self.wchunks else np.zeros(4, np.float32)
        return self.ops, weights, self.peak


def switch_key(*classes):
    """Every class-wide switch of the given classes (their own UPPER_CASE attributes holding a bool / int / float / str /
    tuple / None: FUSE, ROWPAD, PAIR, CHAIN, FOLD_UPSAMPLE, PlanBuilder.X6, ...) as one hashable tuple.  It is part of
    every plan-cache key: a plan records the kernels the switches selected when it was emitted, so flipping one after a
    plan exists must build another plan, not reuse that one."""
    key = []
    for cls in classes:
        for name, v in sorted(vars(cls).items()):
            if name.isupper() and isinstance(v, (bool, int, float, str, tuple, type(None))):
                key.append((cls.__name__, name, v))
    return tuple(key)


class PlanCache:
    """Per-network cache of compiled plans, keyed by batch shape + the emit switches (switch_key).

    * LRU-bounded (``max_plans``): a plan owns an activation arena of several GB at batch >= 1024, so a caller whose
      batch size varies (FacePipeline's 64-row buckets) must not pin one arena per size it has ever seen.
    * The packed weight blob does not depend on the batch size: every plan of one network shares ONE device copy
      (compared by content on the host, so a plan emitted with different fusion switches gets its own).
    ``clear()`` is what load_state_dict / .to() / fuse() call."""

    def __init__(self, max_plans=4):
        self.max_plans = int(max_plans)
        self._plans = {}          # insertion-ordered: oldest first
        self._weights = []        # [(host ndarray, device tensor)]

    def clear(self):
        self._plans = {}
        self._weights = []

    def __len__(self):
        return len(self._plans)

    def __contains__(self, key):
        return key in self._plans

    def get(self, key, build):
        plan = self._plans.pop(key, None)
        if plan is None:
            while len(self._plans) >= self.max_plans:
                self._plans.pop(next(iter(self._plans)))
            plan = build(self)
        self._plans[key] = plan   # most recently used last
        return plan

    def device_weights(self, host, device):
        import torch
        for h, d in self._weights:
            if d.device == device and h.shape == host.shape and np.array_equal(h.view(np.uint32), host.view(np.uint32)):
                return d
        d = torch.from_numpy(host).to(device)
        self._weights.append((host, d))
        return d


class CompiledPlan:
    """Ops + device weights + arena, ready to run on a stream.

    Every tensor a plan exposes (``input``, ``out``, ``r``, ``c``, ``z`` ...) is a VIEW into its arena: the next
    ``run()`` of the same plan overwrites it.  The public forward APIs of the networks return clones; the
    ``*_resident`` variants hand out the views (zero-copy, for callers that consume them before the next run)."""

    def __init__(self, builder, device, cache=None):
        import torch
        ops, weights, arena_floats = builder.finish()
        self.n_ops = len(ops)
        self.N = builder.N            # batch capacity: the arena holds N images; run(n=...) may process fewer
        self.n_run = builder.N
        self.alg_bytes = list(builder.alg_bytes)
        assert len(self.alg_bytes) == self.n_ops
        self.ops = (L.FpOp * max(self.n_ops, 1))(*ops)
        self.device = torch.device(device)
        self.weights = cache.device_weights(weights, self.device) if cache is not None else \
            torch.from_numpy(weights).to(self.device)
        self.arena_floats = int(arena_floats)
        # row-padded buffers rely on pads that nobody ever writes: start from zeros
        alloc = torch.zeros if getattr(builder, "has_rowpad", False) else torch.empty
        self.arena = alloc(self.arena_floats, dtype=torch.float32, device=self.device)
        self.lib = L.load()
        L.check(self.lib.fp_plan_validate(self.ops, self.n_ops, self.weights.numel(), self.arena_floats),
                "fp_plan_validate")

    def buf_tensor(self, buf, N):
        """A torch view [N, H, W, C] of an arena buffer (no copy)."""
        if buf.rowpad:
            return self.arena.as_strided((N, buf.H, buf.W, buf.C), (buf.ns, (buf.W + 1) * buf.C, buf.C, 1), buf.off)
        return self.arena[buf.off: buf.off + N * buf.ns].view(N, buf.H, buf.W, buf.C)

    _timing = None   # (timer, mask) set by bench.py around a timed step; None = plain fp_plan_run
    _ext = None      # ctypes array of fp_ext (external buffers of *_U8 ops), set by set_ext()
    _ext_keep = ()   # the tensors behind it, kept alive

    def set_ext(self, tensors):
        """External device buffers of the plan (e.g. [frames u8, tap tables, LUT] for a *_U8 stem op), in the order the
        ops index them.  The tensors are held until the next set_ext."""
        self._ext_keep = tuple(tensors)
        self._ext = (L.FpExt * max(len(tensors), 1))(*[L.FpExt(t.data_ptr(), t.numel() * t.element_size())
                                                      for t in tensors])

    def set_batch(self, n):
        """Process only the first n <= N images on the following runs.  Every per-image stride of an op is independent
        of the batch size, so this only rewrites the ops' N field; kernel selection (persistent / streaming / per-tile)
        follows the actual n at launch time."""
        n = int(n)
        if not 0 < n <= self.N:
            raise ValueError(f"batch {n} outside the plan's capacity 1..{self.N}")
        if n != self.n_run:
            for i in range(self.n_ops):
                self.ops[i].N = n
            self.n_run = n

    def run(self, n=None):
        if n is not None:
            self.set_batch(n)
        if self._timing is not None:
            return self.run_timed(*self._timing)
        rc = self.lib.fp_plan_run_ext(self.ops, self.n_ops, L.ptr(self.weights), self.weights.numel(),
                                      L.ptr(self.arena), self.arena_floats, self._ext, len(self._ext_keep),
                                      L.current_stream(self.device))
        L.check(rc, "fp_plan_run")

    # ---- measurement support (bench.py) ----
    def new_timer(self):
        t = C.c_void_p()
        L.check(self.lib.fp_timer_create(max(self.n_ops, 1), C.byref(t)), "fp_timer_create")
        return t

    def run_timed(self, timer, mask):
        """Like run(), with HIP events recorded on the stream around the ops selected by mask (bytes, n_ops)."""
        m = (C.c_ubyte * self.n_ops)(*mask)
        rc = self.lib.fp_plan_run_timed_ext(self.ops, self.n_ops, L.ptr(self.weights), self.weights.numel(),
                                            L.ptr(self.arena), self.arena_floats, self._ext, len(self._ext_keep),
                                            L.current_stream(self.device), timer, m)
        L.check(rc, "fp_plan_run_timed")

    def accumulate(self, timer, ms):
        """ms: ctypes float array of n_ops; adds the last timed run's per-op milliseconds."""
        L.check(self.lib.fp_timer_accumulate(timer, ms, self.n_ops), "fp_timer_accumulate")

    def destroy_timer(self, timer):
        self.lib.fp_timer_destroy(timer)

    def kernel_name(self, i):
        """The HIP kernel family op i launches (matches the rocprofv3 kernel-trace names)."""
        return self.lib.fp_op_kernel_name(C.byref(self.ops[i])).decode()

    def compulsory_bytes(self, i, n=None):
        """Bytes op i MUST move for a batch of n images (default: the batch the plan last ran on): every tensor it reads
        once + every tensor it writes once, physical channel counts, weights not counted (they are re-read from L2).
        A fused op is charged for its inputs and outputs only -- the tensors between the reference ops it replaces never
        exist -- so this is the denominator of a physical roofline fraction (bench.py `roofline.frac`); the SURVEY 8(d)
        op-granular figure is algorithmic_bytes()."""
        op = self.ops[i]
        n = self.n_run if n is None else n
        k = op.kind
        if k in (L.OP_STEM_U8, L.OP_YSTEM_U8):
            b_in = op.res_H * op.res_W * 3                       # u8 frame
        else:
            b_in = op.H * op.W * op.Cin * 4
            if op.flags & L.OPF_IN_UP2:                          # the leading res_C channels come from the half-size map
                b_in = op.H * op.W * (op.Cin - op.res_C) * 4 + op.res_H * op.res_W * op.res_C * 4
        cout = op.Cout if k in (L.OP_CONV, L.OP_BLAZEBLOCK, L.OP_DWPW, L.OP_DWBLOCK, L.OP_BLAZEPAIR, L.OP_BLAZECHAIN, L.OP_YSTEM, L.OP_YSTEM_U8,
                                L.OP_STEM_U8, L.OP_SHUFDOWN, L.OP_SHUFUNIT, L.OP_YSTEM2) else op.Cin
        oh, ow = (op.H, op.W) if k in (L.OP_COPY, L.OP_L2NORM) else (op.OH, op.OW)
        b_out = oh * ow * cout * 4 * (2 if op.res_mode == L.RES_SHUFFLE2 else 1)
        b_res = 0
        if k in (L.OP_YSTEM, L.OP_YSTEM_U8):
            b_res = (op.OH // 2) * (op.OW // 2) * op.res_C * 4   # the pooled stem_1 map it also writes
        elif k == L.OP_YSTEM2:
            b_res = op.OH * op.OW * op.res_C * 4                 # the pooled map it reads
        elif (k in (L.OP_CONV, L.OP_DWPW) and op.res_off != op.in_off and
              op.res_mode in (L.RES_ADD_BEFORE_ACT, L.RES_ADD_AFTER_ACT, L.RES_SHUFFLE2)):
            b_res = oh * ow * op.res_C * 4                       # a residual that is not the op's own input (the block
                                                                 # ops' shortcut is their input: read once)
        return n * (b_in + b_out + b_res)

    def flops(self, i, n=None):
        """Arithmetic of op i as the REFERENCE counts it (2 x multiply-accumulates of its convolutions, logical shapes) for a
        batch of n images -- whatever instructions the kernel uses for them."""
        op = self.ops[i]
        n = self.n_run if n is None else n
        k, opix = op.kind, op.OH * op.OW
        if k in (L.OP_CONV, L.OP_STEM_U8):
            f = opix * op.KH * op.KW * op.Cin * op.Cout + (opix * 9 * op.Cout if op.flags & L.OPF_OUT_DW else 0)
        elif k == L.OP_DWCONV:
            f = opix * op.KH * op.KW * op.Cin
        elif k in (L.OP_BLAZEBLOCK, L.OP_DWPW):
            f = opix * (9 * op.Cin + op.Cin * op.Cout)
        elif k == L.OP_BLAZEPAIR and op.stride == 2:
            f = op.H * op.W * (9 * op.Cin + op.Cin * op.Cin) + opix * (9 * op.Cin + op.Cin * op.Cout)
        elif k == L.OP_BLAZEPAIR:
            f = 2 * opix * (9 * op.Cin + op.Cin * op.Cout)
        elif k == L.OP_BLAZECHAIN:
            f = op.Cmid * opix * (9 * op.Cin + op.Cin * op.Cout)
        elif k == L.OP_DWBLOCK:
            f = op.H * op.W * op.Cin * op.Cmid + opix * (9 * op.Cmid + op.Cmid * op.Cout)
        elif k == L.OP_SHUFUNIT:
            f = opix * (2 * op.Cmid * op.Cmid + 9 * op.Cmid)
        elif k == L.OP_YSTEM2:
            f = opix * (9 * op.Cin * op.Cout + 2 * op.Cout * op.Cout)
        elif k == L.OP_SHUFDOWN:   # branch1: dw + 1x1; branch2: 1x1 at full resolution, dw, 1x1
            f = opix * (9 * op.Cin + op.Cin * op.Cmid) + op.H * op.W * op.Cin * op.Cmid + opix * (9 * op.Cmid + op.Cmid * op.Cmid)
        else:
            f = 0
        return 2.0 * n * f

    def bound(self, i):
        """"mfma" for the ops whose kernels run on the matrix cores near their issue limit (the split-MFMA blocks), else "hbm"."""
        return "mfma" if self.ops[i].flags & L.OPF_SPLIT3 else "hbm"

    def algorithmic_bytes(self, i):
        """Op-granular fp32 activation bytes of op i: a conv / linear reads its input once and
        writes its output once (logical channel counts); epilogue-class ops (bias, BN, activation, residual,
        pad, concat, pool, shuffle, upsample, l2norm) are free; a fused op counts the convs it contains."""
        return self.alg_bytes[i]


def validate_on_host(builder):
    """Host-only validation (no GPU): runs fp_plan_validate on the built ops."""
    ops, weights, arena_floats = builder.finish()
    arr = (L.FpOp * max(len(ops), 1))(*ops)
    lib = L.load()
    return lib.fp_plan_validate(arr, len(ops), int(weights.size), int(arena_floats))

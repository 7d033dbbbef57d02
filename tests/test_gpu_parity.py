"""GPU parity tests: HIP path (through the C ABI) vs the oracle and the reference-generated goldens.
Tolerances: bit-exact for indices / keep-sets / cluster membership; 1e-4 absolute for unit-norm embeddings
and cosine scores (BASELINE.json north_star); activations compared with a 1e-4 relative-to-scale bound."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden, rel_err
from face_detection_and_recognition_amd import _lib as L
from face_detection_and_recognition_amd import similarity as S
from face_detection_and_recognition_amd.modules.blazeface.blazeface import (BlazeBlock, BlazeFace, FinalBlazeBlock,
                                                                              generate_anchors)
from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import Depth_Wise, MobileFaceNet
from face_detection_and_recognition_amd.modules.params import npy
from face_detection_and_recognition_amd.plan import CompiledPlan, PlanBuilder
from face_detection_and_recognition_amd.synth import synth_state_dict
from oracle import blazeface_ref, image_ref, mobilefacenet_ref, similarity_ref

pytestmark = pytest.mark.gpu


def plan_rowpad_bufs(net, n):
    """The row-padded buffers of the BlazeFace plan for batch n (re-emitted on the host: same offsets as the cached plan)."""
    pb = net._emit(n)[0]
    pb.finish()
    return pb.rowpad_bufs


def run_block(block, x_nchw, dev, out_hw, cout):
    """Runs one emit()-able block on an NCHW numpy input; returns NCHW numpy."""
    N, C, H, W = x_nchw.shape
    pb = PlanBuilder(N)
    inp = pb.new_buf(H, W, C)
    y = block.emit(pb, inp.view())
    plan = CompiledPlan(pb, dev)
    t = plan.buf_tensor(inp, N)
    t.zero_()
    t[..., :C].copy_(torch.from_numpy(x_nchw).to(dev).permute(0, 2, 3, 1))
    plan.run()
    torch.cuda.synchronize()
    return plan.buf_tensor(y, N)[..., :cout].permute(0, 3, 1, 2).cpu().numpy()


@pytest.mark.parametrize("name,ctor,cout", [("s1", lambda: BlazeBlock(24, 24), 24),
                                             ("s2", lambda: BlazeBlock(24, 48, stride=2), 48),
                                             ("final", lambda: FinalBlazeBlock(96), 96)])
def test_blazeblock_vs_reference_golden(dev, name, ctor, cout):
    g = golden(f"blazeblock_{name}")
    blk = ctor()
    blk.load_state_dict(synth_state_dict(blk.state_dict(), int(g["seed"])))
    y = run_block(blk, g["x"], dev, g["y"].shape[2:], cout)
    assert y.shape == g["y"].shape
    assert rel_err(y, g["y"]) < 1e-5


@pytest.mark.parametrize("back", [True, False])
def test_blazeface_forward_vs_reference_golden(dev, back):
    tag = "back" if back else "front"
    g = golden(f"blazeface_{tag}_forward")
    net = BlazeFace(back)
    net.load_state_dict(synth_state_dict(net.state_dict(), int(g["seed"]), residual_gain=0.5))
    net = net.to(dev)
    x = torch.from_numpy(g["x_u8"]).to(dev)                       # (2, S, S, 3) u8 RGB
    r, c = net.raw_from_u8_nhwc(x)
    torch.cuda.synchronize()
    assert rel_err(r.cpu().numpy(), g["r"]) < 1e-4
    assert rel_err(c.cpu().numpy(), g["c"]) < 1e-4
    # absolute bounds on what the post-processing consumes (a bound relative to the tensor maximum would let a logit error
    # of a few 1e-3 through): scores after clip + sigmoid within 1e-4 (north_star), decoded box / keypoint coordinates
    # (raw / input size, blazeface.py:373-402; anchors have w = h = 1) within 1e-4 of the image
    sig = lambda a: 1.0 / (1.0 + np.exp(-np.clip(a.astype(np.float64), -100.0, 100.0)))
    assert np.abs(sig(c.cpu().numpy()) - sig(g["c"])).max() < 1e-4
    assert np.abs(r.cpu().numpy() - g["r"]).max() / (256.0 if back else 128.0) < 1e-4
    # the float NCHW entry point gives the same numbers
    r2, c2 = net(torch.from_numpy(g["x_u8"]).permute(0, 3, 1, 2).float() / 127.5 - 1.0)
    assert rel_err(r2.cpu().numpy(), g["r"]) < 1e-4


@pytest.mark.parametrize("back", [True, False])
def test_blazeface_fused_and_unfused_plans_agree(dev, back):
    """FP_OP_BLAZEBLOCK (csrc/blaze.hip) against the DWCONV + CONV pair and the reference golden."""
    tag = "back" if back else "front"
    g = golden(f"blazeface_{tag}_forward")
    x = torch.from_numpy(g["x_u8"]).to(dev)
    outs = {}
    for fuse in (True, False):
        BlazeBlock.FUSE = fuse
        try:
            net = BlazeFace(back)
            net.load_state_dict(synth_state_dict(net.state_dict(), int(g["seed"]), residual_gain=0.5))
            net = net.to(dev)
            plan = net.plan_for(2)
            kinds = [plan.ops[i].kind for i in range(plan.n_ops)]
            assert (L.OP_BLAZEBLOCK in kinds) == fuse
            r, c = net.raw_from_u8_nhwc(x)
            torch.cuda.synchronize()
            outs[fuse] = (r.cpu().numpy().copy(), c.cpu().numpy().copy())
        finally:
            BlazeBlock.FUSE = True
    for fuse in (True, False):
        assert rel_err(outs[fuse][0], g["r"]) < 1e-4 and rel_err(outs[fuse][1], g["c"]) < 1e-4
    assert rel_err(outs[True][0], outs[False][0]) < 1e-5


@pytest.mark.parametrize("back,frame_hw,nf", [(True, (576, 1024), 5), (False, (576, 1024), 5), (True, (97, 33), 5),
                                              (True, (300, 211), 5), (False, (1275, 1650), 5),
                                              (True, (576, 1024), 21), (True, (211, 300), 16),
                                              (True, (1024, 576), 17)])   # >= 16 frames: the band stems (fp32 MFMA / split MFMA); the last one a portrait frame: padding left and right
def test_blazeface_letterbox_fused_into_stem_is_bit_exact(dev, back, frame_hw, nf):
    """FP_OP_STEM_U8 (the 5x5 stem resamples the u8 frames through fp_letterbox_tables while it stages its input;
    no fp32 canvas) against the stand-alone letterbox kernel + the fp32 stem: identical raw network outputs."""
    from face_detection_and_recognition_amd.modules.blazeface import blazeface as B
    from face_detection_and_recognition_amd.modules.blazeface.model import BlazeFaceModel
    rng = np.random.default_rng(frame_hw[1])
    frames = torch.from_numpy(rng.integers(0, 256, (nf,) + frame_hw + (3,), dtype=np.uint8)).to(dev)
    net = BlazeFace(back)
    net.load_state_dict(synth_state_dict(net.state_dict(), 100 + int(back), residual_gain=0.5))
    net = net.to(dev)
    net.set_anchors(generate_anchors(back))
    model = BlazeFaceModel("", 0.7, 0.12, "back" if back else "front", device=str(dev), net=net)
    outs = {}
    stem_x6 = B.BlazeFace.STEM_X6
    for flag in (True, False, "x6"):
        B.BlazeFace.FUSE_LETTERBOX = bool(flag)
        B.BlazeFace.STEM_X6 = flag == "x6"       # the fp32 stems are the bit-identical ones; the split-MFMA band stem below
        try:
            model.raw_batch(frames)
            torch.cuda.synchronize()
            assert (net.last_plan.input is None) == bool(flag)
            if flag:
                band = "stem5_u8_x6_kernel" if flag == "x6" else "stem5_u8_band_kernel"
                assert net.last_plan.kernel_name(0) == (band if back and nf >= 16 else "stem_conv_kernel<5, 1, true>")
            outs[flag] = (net.last_plan.r.clone().cpu().numpy(), net.last_plan.c.clone().cpu().numpy())
        finally:
            B.BlazeFace.FUSE_LETTERBOX = True
            B.BlazeFace.STEM_X6 = stem_x6
    np.testing.assert_array_equal(outs[True][0], outs[False][0])
    np.testing.assert_array_equal(outs[True][1], outs[False][1])
    # FP_OPF_SPLIT3 on the band stem (stem5_u8_x6_kernel: the conv on the bf16 matrix cores, fp32-equivalent split arithmetic):
    # the same raw outputs to fp32 rounding through the whole network
    assert rel_err(outs["x6"][0], outs[False][0]) < 1e-5 and rel_err(outs["x6"][1], outs[False][1]) < 1e-5


def test_u8_stems_refuse_tap_tables_of_another_geometry(dev, lib):
    """The *_U8 stems address the frames through the tap tables; tables built for a LARGER frame would send their 8-byte
    window loads past the frames buffer.  fp_letterbox_tables records the geometry in the tables' last entry and the
    kernels compare it with the op (device side, per workgroup): a mismatch writes nothing and faults nothing."""
    from face_detection_and_recognition_amd.modules.blazeface.model import BlazeFaceModel
    from face_detection_and_recognition_amd.modules.utils.image import letterbox_geometry
    rng = np.random.default_rng(3)
    fh, fw = 120, 200
    frames = torch.from_numpy(rng.integers(0, 256, (3, fh, fw, 3), dtype=np.uint8)).to(dev)
    net = BlazeFace(True)
    net.load_state_dict(synth_state_dict(net.state_dict(), 101, residual_gain=0.5))
    net = net.to(dev)
    net.set_anchors(generate_anchors(True))
    model = BlazeFaceModel("", 0.7, 0.12, "back", device=str(dev), net=net)
    model.raw_batch(frames)
    torch.cuda.synchronize()
    plan = net.last_plan
    good = plan.r.clone()
    tables = plan.tables[1]
    keep = tables.clone()
    # overwrite with tables for 1080 x 1920 frames (same canvas): every tap offset points far outside the 120 x 200 frames
    sw, sh, left, top = letterbox_geometry(1920, 1080, 256, 256)
    L.check(lib.fp_letterbox_tables(1080, 1920, 256, 256, 0, 0, 1920, 1080, left, top, sw, sh, 125, 1, L.ptr(tables),
                                    L.current_stream(dev)), "tables")
    plan.arena.fill_(-7.0)
    L.check(lib.fp_plan_run_ext(plan.ops, 1, L.ptr(plan.weights), plan.weights.numel(), L.ptr(plan.arena),   # the stem alone
                                plan.arena_floats, plan._ext, len(plan._ext_keep), L.current_stream(dev)), "stem only")
    torch.cuda.synchronize()
    assert float((plan.arena + 7.0).abs().max()) == 0.0     # the stem wrote nothing, anywhere
    tables.copy_(keep)
    plan.arena.zero_()                                     # row-padded buffers rely on zero pads
    plan.run()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(plan.r.cpu().numpy(), good.cpu().numpy())


@pytest.mark.parametrize("back,n", [(True, 2), (True, 19), (False, 5), (True, 40)])
def test_blazeface_row_padded_chain_matches_dense(dev, back, n):
    """The 24 -> 24 stride-1 blocks on the wave-private kernel (csrc/blazewp.hip blazeblock_wp_kernel, row-padded
    activations: include/facepath.h FP_OPF_*) against the same network with dense activations: the stem and the
    stride-2 block write the padded layout (per-tile kernel at n = 2, persistent kernel at n >= 16 on the 128 x 128
    map), the last block of a chain writes dense again.  Same taps, same k order: the outputs agree to fp32 rounding
    of one reassociated sum at most."""
    rng = np.random.default_rng(n)
    S = 256 if back else 128
    x = torch.from_numpy(rng.integers(0, 256, (n, S, S, 3), dtype=np.uint8)).to(dev)
    outs = {}
    for flag in (False, True):   # the row-padded network last: `net` below is that one
        BlazeBlock.ROWPAD = flag
        try:
            net = BlazeFace(back)
            net.load_state_dict(synth_state_dict(net.state_dict(), 7, residual_gain=0.5))
            net = net.to(dev)
            plan = net.plan_for(n)
            names = [plan.kernel_name(i) for i in range(plan.n_ops)]
            assert any(nm.startswith(("blazeblock_wp_kernel", "blazepair")) for nm in names) == flag     # the row-padded kernels
            r, c = net.raw_from_u8_nhwc(x)
            torch.cuda.synchronize()
            outs[flag] = (r.cpu().numpy().copy(), c.cpu().numpy().copy())
        finally:
            BlazeBlock.ROWPAD = True
    assert rel_err(outs[True][0], outs[False][0]) < 2e-6
    assert rel_err(outs[True][1], outs[False][1]) < 2e-6
    # a second run on the same plan: the pads are still zero (nobody writes them) -- checked on the result and directly:
    # zeroing every row-padded buffer's interior must leave its whole region zero
    r2, _ = net.raw_from_u8_nhwc(x)
    assert rel_err(r2.cpu().numpy(), outs[False][0]) < 2e-6
    plan = net.plan_for(n)
    regions = sorted({(b.off - (b.W + 2) * b.C, n * b.ns, b.H, b.W, b.C, b.off) for b in plan_rowpad_bufs(net, n)})
    assert regions
    for base, size, H, W, C, off in regions:
        plan.arena.as_strided((n, H, W, C), (size // n, (W + 1) * C, C, 1), off).zero_()
        assert float(plan.arena[base: base + size].abs().max()) == 0.0


@pytest.mark.parametrize("stride,in_rp,out_rp,n,hw,c", [
    (1, True, True, 3, (64, 64), 24),      # wave-private kernel, row-padded -> row-padded
    (1, True, False, 3, (64, 96), 24),     # ... -> dense (last block of a chain), 3 strips
    (1, True, True, 5, (8, 128), 24),      # two bands only, 4 strips
    (2, False, True, 3, (128, 128), 24),   # stride-2 block writes the padded layout: per-tile kernel
    (2, False, True, 40, (128, 128), 24),  # ... persistent kernel (>= 2048 tiles)
    (1, False, True, 3, (64, 64), 24),     # dense -> row-padded on the stride-1 per-tile kernel
    (1, True, True, 5, (32, 32), 48),      # small-map kernel: one row per tile
    (1, True, False, 7, (32, 32), 48),
    (1, True, True, 5, (16, 16), 96),      # two rows per tile, 4 channel passes
    (1, True, False, 3, (6, 16), 96),      # three tiles per image
    (1, False, False, 5, (16, 16), 96),    # dense input: emit() inserts the copy into the padded layout
])
def test_blazeblock_row_padded_layouts_vs_oracle(dev, stride, in_rp, out_rp, n, hw, c):
    """One c -> c BlazeBlock with its input and / or output in the row-padded layout (include/facepath.h FP_OPF_*)
    against the CPU oracle (blazeface.py:12-47), and the pads of the output buffer still zero afterwards."""
    rng = np.random.default_rng(n + stride)
    H, W = hw
    blk = BlazeBlock(c, c, stride=stride)
    sd = synth_state_dict(blk.state_dict(), 77 + stride)
    blk.load_state_dict(sd)
    x = rng.normal(0, 1, (n, c, H, W)).astype(np.float32)
    pb = PlanBuilder(n)
    inp = (pb.new_buf_rowpad if in_rp else pb.new_buf)(H, W, c)
    y = blk.emit(pb, inp.view(), out_rowpad=out_rp)
    assert y.rowpad == out_rp
    plan = CompiledPlan(pb, dev)
    name = plan.kernel_name(plan.n_ops - 1)
    assert name.startswith("blazeblock_wp_kernel" if c == 24 else "blazeblock_wps_kernel") == (in_rp or c != 24), name
    plan.buf_tensor(inp, n).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    for _ in range(2):
        plan.run()
    torch.cuda.synchronize()
    got = plan.buf_tensor(y, n).permute(0, 3, 1, 2).cpu().numpy()
    ref = blazeface_ref._blaze_block(dict(sd), "", torch.from_numpy(x), stride).numpy()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-5
    if out_rp:   # everything in the output region that is not an interior pixel is still zero
        OH, OW = y.H, y.W
        base = y.off - (OW + 2) * y.C
        region = plan.arena[base: base + n * y.ns].clone()
        plan.buf_tensor(y, n).zero_()
        left = plan.arena[base: base + n * y.ns]
        assert float(region.abs().sum()) > 0 and float(left.abs().max()) == 0.0


@pytest.mark.parametrize("hw,n,out_rp", [
    ((128, 128), 3, True),     # four strips per row, 8-row bands (few images): every band boundary case
    ((128, 128), 20, False),   # dense output
    ((64, 64), 5, True),       # two bands per workgroup, odd band count: the last workgroup's second half is idle
    ((64, 64), 16, False),
    ((64, 128), 2, True),      # non-square map
])
def test_blazepair_two_blocks_in_one_kernel_vs_oracle(dev, hw, n, out_rp):
    """FP_OP_BLAZEPAIR (csrc/blazepair.hip): two stride-1 24 -> 24 BlazeBlocks, the tensor between them kept in an LDS
    ring, against blazeface_ref._blaze_block applied twice (torch fp32 on the CPU; blazeface.py:12-47) and against the
    two single-block launches on the same input (same arithmetic per block: 2e-6 of scale)."""
    H, W = hw
    rng = np.random.default_rng(500 + H + n)
    blks = [BlazeBlock(24, 24), BlazeBlock(24, 24)]
    sds = []
    for k, b in enumerate(blks):
        sd = synth_state_dict(b.state_dict(), 1500 + k)
        b.load_state_dict(sd)
        sds.append(sd)
    x = rng.normal(0, 1, (n, 24, H, W)).astype(np.float32)
    outs = {}
    for pair in (True, False):
        pb = PlanBuilder(n)
        inp = pb.new_buf_rowpad(H, W, 24)
        if pair:
            assert blks[0].pairs_with(blks[1], pb, inp.view())
            y = blks[0].emit_pair(blks[1], pb, inp.view(), out_rowpad=out_rp)
        else:
            mid = blks[0].emit(pb, inp.view(), out_rowpad=True)
            y = blks[1].emit(pb, mid.view(), out_rowpad=out_rp)
        plan = CompiledPlan(pb, dev)
        names = [plan.kernel_name(i) for i in range(plan.n_ops)]
        assert names == (["blazepair_kernel<%d>" % W] if pair else ["blazeblock_wp_kernel<24, 4>"] * 2), names
        plan.buf_tensor(inp, n).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
        yt = plan.buf_tensor(y, n)
        yt.fill_(float("nan"))
        plan.run()
        torch.cuda.synchronize()
        outs[pair] = yt.permute(0, 3, 1, 2).cpu().numpy()
        if out_rp:          # the pads of a row-padded output stay zero (the next block's window reads them)
            full = plan.arena[y.off - (W + 2) * 24: y.off - (W + 2) * 24 + n * y.ns].clone()
            yt.zero_()
            assert float(plan.arena[y.off - (W + 2) * 24: y.off - (W + 2) * 24 + n * y.ns].abs().max()) == 0.0
            assert float(full.abs().sum()) > 0
    t = torch.from_numpy(x)
    for sd in sds:
        t = blazeface_ref._blaze_block(sd, "", t, 1)
    ref = t.numpy()
    assert outs[True].shape == ref.shape and np.isfinite(outs[True]).all()
    assert rel_err(outs[True], ref) < 1e-5
    np.testing.assert_allclose(outs[True], ref, rtol=1e-5, atol=2e-5)
    assert rel_err(outs[True], outs[False]) < 2e-6


@pytest.mark.parametrize("hw,cout2,n,out_rp", [
    ((128, 128), 24, 3, True),      # BlazeFace-back's first stage end: four strips, two of them make the 64-pixel output rows
    ((128, 128), 24, 18, False),    # dense output, several bands per image and images per launch
    ((64, 64), 48, 5, True),        # second stage end: 24 -> 48 (two 32-column halves, channels >= 24 without shortcut), odd band count
    ((64, 64), 48, 16, False),
    ((64, 128), 48, 2, True),       # non-square map
    ((32, 64), 24, 3, False),
    ((16, 64), 24, 1, True),        # the smallest map: two bands of four output rows, one image
    ((40, 128), 48, 1, False),      # 20 output rows = five bands of four
])
def test_blazepair_s2_stride1_plus_stride2_block_vs_oracle(dev, hw, cout2, n, out_rp):
    """FP_OP_BLAZEPAIR with stride = 2 (csrc/blazepairs2.hip): a stride-1 24 -> 24 BlazeBlock and the stride-2 block behind it
    (24 -> 24 / 24 -> 48: F.pad(0, 2, 0, 2), depthwise stride 2, shortcut = channel-padded 2 x 2 max pool) in one kernel, the
    full-size tensor between them in an LDS ring -- against blazeface_ref._blaze_block applied twice (torch fp32 on the CPU;
    blazeface.py:12-47) and against the two separate launches (same arithmetic per block: 2e-6 of scale).  The last output
    row / column read the zero row / pixels of the pad."""
    H, W = hw
    rng = np.random.default_rng(900 + H + n + cout2)
    blks = [BlazeBlock(24, 24), BlazeBlock(24, cout2, stride=2)]
    sds = []
    for k, b in enumerate(blks):
        sd = synth_state_dict(b.state_dict(), 1700 + k)
        b.load_state_dict(sd)
        sds.append(sd)
    x = rng.normal(0, 1, (n, 24, H, W)).astype(np.float32)
    x[:, :, -1, :] += 1.5         # bottom row / right column distinct: they meet the pad of the stride-2 window
    x[:, :, :, -1] -= 1.0
    outs = {}
    for fused in (True, False):
        pb = PlanBuilder(n)
        inp = pb.new_buf_rowpad(H, W, 24)
        if fused:
            assert blks[0].pairs_with_s2(blks[1], pb, inp.view())
            y = blks[0].emit_pair_s2(blks[1], pb, inp.view(), out_rowpad=out_rp)
        else:
            mid = blks[0].emit(pb, inp.view(), out_rowpad=False)
            y = blks[1].emit(pb, mid.view(), out_rowpad=out_rp and blks[1].fused(pb, mid.view()))
        plan = CompiledPlan(pb, dev)
        names = [plan.kernel_name(i) for i in range(plan.n_ops)]
        if fused:
            assert names == ["blazepair_s2_kernel<%d, %d>" % (W, cout2)], names
        plan.buf_tensor(inp, n).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
        yt = plan.buf_tensor(y, n)
        yt.fill_(float("nan"))
        for _ in range(2):
            plan.run()
        torch.cuda.synchronize()
        outs[fused] = yt.permute(0, 3, 1, 2).cpu().numpy()
        if fused and out_rp:        # the pads of a row-padded output stay zero (the next block's window reads them)
            OW = W // 2
            full = plan.arena[y.off - (OW + 2) * cout2: y.off - (OW + 2) * cout2 + n * y.ns].clone()
            yt.zero_()
            assert float(plan.arena[y.off - (OW + 2) * cout2: y.off - (OW + 2) * cout2 + n * y.ns].abs().max()) == 0.0
            assert float(full.abs().sum()) > 0
    t = blazeface_ref._blaze_block(sds[0], "", torch.from_numpy(x), 1)
    ref = blazeface_ref._blaze_block(sds[1], "", t, 2).numpy()
    assert outs[True].shape == ref.shape == (n, cout2, H // 2, W // 2) and np.isfinite(outs[True]).all()
    assert rel_err(outs[True], ref) < 1e-5
    np.testing.assert_allclose(outs[True], ref, rtol=1e-5, atol=2e-5)
    assert rel_err(outs[True], outs[False]) < 2e-6


@pytest.mark.parametrize("nblk,n", [(7, 5), (2, 3), (1, 2), (7, 260)])
def test_blazechain_run_of_blocks_in_one_kernel_vs_oracle(dev, nblk, n):
    """FP_OP_BLAZECHAIN (csrc/blazechain.hip): nblk stride-1 96 -> 96 BlazeBlocks on the 16 x 16 map, the image kept in LDS
    for the whole run (depthwise through DPP row shifts, the 1x1 convs as bf16x6 split MFMAs), against
    blazeface_ref._blaze_block applied nblk times (torch fp32 on the CPU; blazeface.py:12-47,146-152) and against the
    single-block launches (fp32 MFMA) on the same input.  Inputs with structure at the image border (the zero padding
    comes from DPP bound_ctrl and the two zero rows in LDS): a constant image would hide a wrong neighbour."""
    rng = np.random.default_rng(4000 + nblk + n)
    blks = [BlazeBlock(96, 96) for _ in range(nblk)]
    sds = []
    for k, b in enumerate(blks):
        sd = synth_state_dict(b.state_dict(), 2500 + k)
        b.load_state_dict(sd)
        sds.append(sd)
    x = rng.normal(0, 1, (n, 96, 16, 16)).astype(np.float32)
    x[:, :, :, 0] += 2.0          # left column, top row: distinct from the interior
    x[:, :, 0, :] -= 1.5
    outs = {}
    for chain in (True, False):
        pb = PlanBuilder(n)
        inp = pb.new_buf(16, 16, 96)
        if chain:
            assert pb.blazechain_supported(inp.view()) and all(b.chains() for b in blks)
            y = pb.new_buf(16, 16, 96)
            pb.blazechain(inp.view(), [(npy(b.convs[0].weight), npy(b.convs[0].bias), npy(b.convs[1].weight), npy(b.convs[1].bias))
                                       for b in blks], y.view())
        else:
            y = inp
            for b in blks:
                y = b.emit(pb, y.view())
        plan = CompiledPlan(pb, dev)
        names = [plan.kernel_name(i) for i in range(plan.n_ops)]
        if chain:
            assert names == ["blazechain96_kernel"], names
        else:
            assert names.count("blazeblock_wps_kernel<96>") == nblk, names
        plan.buf_tensor(inp, n).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
        yt = plan.buf_tensor(y, n)
        if chain:
            yt.fill_(float("nan"))
        for _ in range(2):        # the op must not have modified its input
            plan.run()
        torch.cuda.synchronize()
        outs[chain] = yt.permute(0, 3, 1, 2).cpu().numpy()
    t = torch.from_numpy(x[:8])
    for sd in sds:
        t = blazeface_ref._blaze_block(sd, "", t, 1)
    ref = t.numpy()
    assert outs[True].shape == x.shape and np.isfinite(outs[True]).all()
    assert rel_err(outs[True][:8], ref) < 1e-5
    np.testing.assert_allclose(outs[True][:8], ref, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()))
    assert rel_err(outs[True], outs[False]) < 5e-6
    # in place (include/facepath.h BLAZECHAIN: in and out may be the same view): a workgroup reads its whole image first
    pb = PlanBuilder(n)
    inp = pb.new_buf(16, 16, 96)
    pb.blazechain(inp.view(), [(npy(b.convs[0].weight), npy(b.convs[0].bias), npy(b.convs[1].weight), npy(b.convs[1].bias))
                               for b in blks], inp.view())
    plan = CompiledPlan(pb, dev)
    plan.buf_tensor(inp, n).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    plan.run()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(plan.buf_tensor(inp, n).permute(0, 3, 1, 2).cpu().numpy(), outs[True])


def test_blazeblock_fused_ragged_tail(dev):
    """M = N*OH*OW not a multiple of the 128-row tile, odd batch: the tail tile must not write out of range."""
    rng = np.random.default_rng(12)
    for stride, cin, cout, hw in ((1, 24, 24, 10), (2, 24, 48, 10), (1, 48, 48, 6), (1, 28, 32, 6)):
        blk = BlazeBlock(cin, cout, stride=stride)
        sd = synth_state_dict(blk.state_dict(), 900 + cin + stride)
        blk.load_state_dict(sd)
        x = rng.normal(0, 1, (3, cin, hw, hw)).astype(np.float32)
        y = run_block(blk, x, dev, None, cout)
        xt = torch.from_numpy(x)
        ref = blazeface_ref._blaze_block({k: v for k, v in sd.items()}, "", xt, stride).numpy()
        assert y.shape == ref.shape
        assert rel_err(y, ref) < 1e-5


def test_blazeface_decode_vs_reference_golden(dev, lib):
    g = golden("blazeface_decode")
    B, A = g["raw_box"].shape[:2]
    rb = torch.from_numpy(g["raw_box"]).to(dev)
    rs = torch.from_numpy(g["raw_score"]).to(dev)
    an = torch.from_numpy(g["anchors"]).to(dev)
    cand = torch.zeros((B, A, 17), device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    L.check(lib.fp_blaze_decode(L.ptr(rb), L.ptr(rs), L.ptr(an), B, A, 256.0, 256.0, 256.0, 256.0, 100.0, 0.65,
                                L.ptr(cand), L.ptr(cnt), None), "decode")
    torch.cuda.synchronize()
    assert cnt.cpu().numpy().tolist() == g["counts"].tolist()       # same candidate sets
    for i in range(B):
        d = g[f"dets{i}"]
        got = cand[i, :len(d)].cpu().numpy()
        np.testing.assert_array_equal(got[:, :16], d[:, :16])      # decode is bit-exact (fp-contract off)
        np.testing.assert_allclose(got[:, 16], d[:, 16], rtol=0, atol=2e-7)


def test_blazeface_decode_extreme_scores_and_threshold_edge(dev, lib):
    """Raw scores far outside the clip range, infinities, NaN and values whose sigmoid sits next to min_score: the candidate set
    is the oracle's (blazeface.py:321-371: clamp to +-100, sigmoid, `>= min_score`; a NaN score is not a candidate), rows
    decoded exactly, scores 2e-7."""
    g = golden("blazeface_decode")
    A = g["raw_box"].shape[1]
    rng = np.random.default_rng(3)
    B = 3
    rb = np.tile(g["raw_box"][:1], (B, 1, 1)).astype(np.float32)
    edge = np.float32(np.log(0.65 / 0.35))                                  # sigmoid(edge) ~ 0.65
    pool = np.array([-1e30, -1000, -100.5, -100, -1, 0, 1, 100, 100.5, 1000, 1e30, np.inf, -np.inf, np.nan,
                     edge, np.nextafter(edge, np.float32(9)), np.nextafter(edge, np.float32(-9)), edge + 1e-6, edge - 1e-6],
                    np.float32)
    rs = pool[rng.integers(0, len(pool), (B, A, 1))]
    rbt, rst, an = (torch.from_numpy(a).to(dev) for a in (rb, rs, g["anchors"]))
    cand = torch.zeros((B, A, 17), device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    L.check(lib.fp_blaze_decode(L.ptr(rbt), L.ptr(rst), L.ptr(an), B, A, 256.0, 256.0, 256.0, 256.0, 100.0, 0.65,
                                L.ptr(cand), L.ptr(cnt), None), "decode")
    torch.cuda.synchronize()
    ref = blazeface_ref.tensors_to_detections(torch.from_numpy(rb), torch.from_numpy(rs), torch.from_numpy(g["anchors"]), 256.0)
    assert cnt.cpu().numpy().tolist() == [len(d) for d in ref]
    for i, d in enumerate(ref):
        got = cand[i, :len(d)].cpu().numpy()
        assert np.isfinite(got).all()
        np.testing.assert_array_equal(got[:, :16], d.numpy()[:, :16])
        np.testing.assert_allclose(got[:, 16], d.numpy()[:, 16], rtol=0, atol=2e-7)


def _wnms_gpu(dev, lib, dets_list, thr=0.3):
    B = len(dets_list)
    nmax = 896
    d = torch.zeros((B, nmax, 17), device=dev)
    cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    for i, a in enumerate(dets_list):
        d[i, :len(a)] = torch.from_numpy(a).to(dev)
        cnt[i] = len(a)
    out = torch.zeros((B, nmax, 17), device=dev)
    oc = torch.zeros((B,), dtype=torch.int32, device=dev)
    mem = torch.full((B, nmax), -1, dtype=torch.int32, device=dev)
    L.check(lib.fp_blaze_weighted_nms(L.ptr(d), L.ptr(cnt), B, nmax, thr, L.ptr(out), L.ptr(oc), L.ptr(mem), None),
            "wnms")
    torch.cuda.synchronize()
    return out.cpu().numpy(), oc.cpu().numpy(), mem.cpu().numpy()


def test_blazeface_weighted_nms_vs_reference_golden(dev, lib):
    g = golden("blazeface_wnms")
    names = ["no_overlap", "clusters", "chains", "all_overlap", "many", "single"]
    ins = [g[n + "_in"] for n in names]
    out, oc, mem = _wnms_gpu(dev, lib, ins + [np.zeros((0, 17), np.float32)])
    assert oc[-1] == 0                                              # empty image
    for i, n in enumerate(names):
        ref_out = g[n + "_out"]
        assert oc[i] == len(ref_out), n                             # same number of faces
        _, ref_mem = blazeface_ref.weighted_nms(ins[i], 0.3)
        np.testing.assert_array_equal(mem[i, :len(ins[i])], ref_mem.numpy())   # bit-exact cluster membership
        np.testing.assert_allclose(out[i, :oc[i]], ref_out, rtol=0, atol=1e-5)


def test_blazeface_private_postprocessing_names_vs_reference_golden(dev):
    """The reference's BlazeFace exposes _decode_boxes / _tensors_to_detections / _weighted_non_max_suppression
    (blazeface.py:373, :321, :404); the same names here run the HIP kernels and reproduce the reference's own outputs
    (tests/golden/blazeface_decode.npz, blazeface_wnms.npz): decoded boxes exact, candidate lists exact (scores 2e-7),
    blended detections 1e-5, torch and numpy return forms."""
    g = golden("blazeface_decode")
    net = BlazeFace(True).to(dev)
    boxes = net._decode_boxes(torch.from_numpy(g["raw_box"]), torch.from_numpy(g["anchors"]))
    np.testing.assert_array_equal(boxes.cpu().numpy(), g["boxes"])
    assert isinstance(net._decode_boxes(g["raw_box"], g["anchors"], use_numpy=True), np.ndarray)
    dets = net._tensors_to_detections(torch.from_numpy(g["raw_box"]), torch.from_numpy(g["raw_score"]),
                                      torch.from_numpy(g["anchors"]))
    assert [len(d) for d in dets] == g["counts"].tolist()
    for i, d in enumerate(dets):
        np.testing.assert_array_equal(d.cpu().numpy()[:, :16], g[f"dets{i}"][:, :16])
        np.testing.assert_allclose(d.cpu().numpy()[:, 16], g[f"dets{i}"][:, 16], rtol=0, atol=2e-7)
    w = golden("blazeface_wnms")
    for name in ("no_overlap", "clusters", "chains", "all_overlap", "many", "single"):
        faces = net._weighted_non_max_suppression(torch.from_numpy(w[name + "_in"]))
        assert len(faces) == len(w[name + "_out"]) and all(f.shape == (17,) for f in faces)
        np.testing.assert_allclose(torch.stack(faces).cpu().numpy(), w[name + "_out"], rtol=0, atol=1e-5)
    assert net._weighted_non_max_suppression(torch.zeros((0, 17))) == []
    faces = net._weighted_non_max_suppression(w["clusters_in"], use_numpy=True)
    assert isinstance(faces[0], np.ndarray) and len(faces) == len(w["clusters_out"])
    with pytest.raises(AssertionError):
        net._tensors_to_detections(torch.zeros((1, 10, 16)), torch.zeros((1, 10, 1)), torch.zeros((10, 4)))


def test_blazeface_weighted_nms_degenerate_box_terminates(dev, lib):
    # ymin > ymax makes the reference loop forever; the kernel emits the box alone and goes on.
    d = np.array([[0.5, 0.5, 0.4, 0.6] + [0.1] * 12 + [0.9], [0.1, 0.1, 0.3, 0.3] + [0.2] * 12 + [0.8]], np.float32)
    out, oc, mem = _wnms_gpu(dev, lib, [d])
    assert oc[0] == 2 and mem[0, :2].tolist() == [0, 1]
    np.testing.assert_array_equal(out[0, :2], d)


def test_blazeface_weighted_nms_full_size_properties(dev, lib):
    # 256 images x up to 896 candidates (BASELINE config 2 post-processing size): membership is a partition,
    # counts match the oracle on a sample, output scores are descending in "first" order only per cluster.
    rng = np.random.default_rng(5)
    ins = []
    for i in range(256):
        n = int(rng.integers(0, 897)) if i % 7 else 896
        xy = rng.uniform(0, 0.8, (n, 2)); wh = rng.uniform(0.05, 0.3, (n, 2))
        ins.append(np.concatenate([xy, xy + wh, rng.uniform(0, 1, (n, 12)), rng.uniform(0.65, 1, (n, 1))], 1).astype(np.float32))
    out, oc, mem = _wnms_gpu(dev, lib, ins)
    for i in (0, 1, 5, 100, 255):
        ref_out, ref_mem = blazeface_ref.weighted_nms(ins[i], 0.3)
        assert oc[i] == len(ref_out)
        np.testing.assert_array_equal(mem[i, :len(ins[i])], ref_mem.numpy())
        np.testing.assert_allclose(out[i, :oc[i]], ref_out.numpy(), rtol=0, atol=2e-5)
    for i in range(256):
        m = mem[i, :len(ins[i])]
        if len(m):
            assert m.min() >= 0 and m.max() == oc[i] - 1 and len(np.unique(m)) == oc[i]


def test_blazeface_weighted_nms_grid_quantised_boxes_vs_oracle(dev, lib):
    """Adversarial geometry for the clustering: corners on a 1/8 grid, so that many boxes coincide exactly, touch along
    an edge (IoU 0) or nest, IoUs are small exact rationals and whole groups share the SAME overlap with the cluster head; scores
    distinct but only 2^-12 apart (the order is defined, the blend weights nearly equal).  120 images of 1..48 candidates,
    three thresholds: membership bit-exact, blended rows 2e-5."""
    rng = np.random.default_rng(77)
    for thr in (0.3, 0.5, 0.125):
        ins = []
        for i in range(120):
            n = int(rng.integers(1, 49))
            x0 = rng.integers(0, 6, n) / 8.0
            y0 = rng.integers(0, 6, n) / 8.0
            w = rng.integers(1, 4, n) / 8.0
            h = rng.integers(1, 4, n) / 8.0
            score = 0.7 + rng.permutation(n) / 4096.0                 # distinct, close
            kp = rng.integers(0, 9, (n, 12)) / 8.0
            ins.append(np.concatenate([np.stack([y0, x0, y0 + h, x0 + w], 1), kp, score[:, None]], 1).astype(np.float32))
        out, oc, mem = _wnms_gpu(dev, lib, ins, thr=thr)
        for i, a in enumerate(ins):
            ref_out, ref_mem = blazeface_ref.weighted_nms(a, thr)
            assert oc[i] == len(ref_out), (thr, i)
            np.testing.assert_array_equal(mem[i, :len(a)], ref_mem.numpy(), err_msg=str((thr, i)))
            np.testing.assert_allclose(out[i, :oc[i]], ref_out.numpy(), rtol=0, atol=2e-5)


def test_mobilefacenet_depthwise_vs_reference_golden(dev):
    g = golden("mobilefacenet_depthwise")
    a = Depth_Wise(64, 64, residual=True, kernel=(3, 3), stride=(1, 1), padding=(1, 1), groups=128)
    a.load_state_dict(synth_state_dict(a.state_dict(), 301))
    b = Depth_Wise(64, 128, kernel=(3, 3), stride=(2, 2), padding=(1, 1), groups=256)
    b.load_state_dict(synth_state_dict(b.state_dict(), 302))
    ya = run_block(a, g["x"], dev, (14, 14), 64)
    yb = run_block(b, g["x"], dev, (7, 7), 128)
    assert rel_err(ya, g["y_res"]) < 1e-5
    assert rel_err(yb, g["y_down"]) < 1e-5


def test_mobilefacenet_fused_and_unfused_plans_agree(dev):
    """FP_OP_DWPW (csrc/dwpw.hip: conv_dw + project fused) against the DWCONV + CONV pair and the golden,
    plus block-level checks at every map size of the network (28: P=4, 14: P=2, 7: P=1; stride 1 and 2)."""
    g = golden("mobilefacenet_forward")
    outs = {}
    for fuse in (True, False):
        Depth_Wise.FUSE = fuse
        try:
            net = MobileFaceNet(512)
            net.load_state_dict(synth_state_dict(net.state_dict(), int(g["seed"])))
            net = net.to(dev)
            plan = net.plan_for(4)
            kinds = [plan.ops[i].kind for i in range(plan.n_ops)]
            assert (L.OP_DWPW in kinds or L.OP_DWBLOCK in kinds) == fuse   # fused forms: dw->pw kernels / whole-block kernels
            outs[fuse] = net(torch.from_numpy(g["x"])).cpu().numpy().copy()
        finally:
            Depth_Wise.FUSE = True
    assert np.abs(outs[True] - g["emb"]).max() < 1e-4 and np.abs(outs[False] - g["emb"]).max() < 1e-4
    assert np.abs(outs[True] - outs[False]).max() < 2e-6
    rng = np.random.default_rng(21)
    for cin, cout, groups, stride, hw, residual in ((64, 64, 128, 1, 28, True), (64, 128, 256, 2, 28, False),
                                                    (128, 128, 256, 1, 14, True), (128, 128, 512, 2, 14, False),
                                                    (128, 128, 256, 1, 7, True), (64, 64, 128, 2, 56, False)):
        blk = Depth_Wise(cin, cout, residual=residual, kernel=(3, 3), stride=(stride, stride), padding=(1, 1),
                         groups=groups)
        sd = synth_state_dict(blk.state_dict(), 700 + groups + stride + hw)
        blk.load_state_dict(sd)
        x = rng.normal(0, 1, (3, cin, hw, hw)).astype(np.float32)
        y = run_block(blk, x, dev, None, cout)
        ref = mobilefacenet_ref._depth_wise({k: torch.as_tensor(v) for k, v in sd.items()}, "", torch.from_numpy(x),
                                            stride, residual).numpy()
        assert y.shape == ref.shape
        assert rel_err(y, ref) < 1e-5, (cin, cout, groups, stride, hw)


@pytest.mark.parametrize("cin,cout,groups,stride,hw,residual,n", [
    (64, 64, 128, 1, 28, True, 176),     # dwpw_wp<2,1> + pws<64>, whole tiles
    (64, 64, 128, 1, 28, True, 169),     # partial last patch tile; M % 32 != 0 -> conv_igemm for the expand conv
    (64, 64, 128, 2, 56, False, 168),    # dwpw_wp<2,2>
    (128, 128, 256, 1, 14, True, 672),   # dwpw_persist<4,1> + pws<128>
    (128, 128, 256, 1, 14, False, 379),  # ... without residual, ragged last tile
])
def test_mobilefacenet_persistent_kernels_large_batch(dev, cin, cout, groups, stride, hw, residual, n):
    """The persistent / streaming kernels (csrc/dwpw.hip dwpw_persist_kernel, csrc/pws.hip) only take over above
    ~1e5 output pixels, more than the golden fixtures hold: Depth_Wise blocks at bench-like batch sizes against the
    oracle (mobilefacenet_ref._depth_wise = torch fp32 on the CPU), incl. a ragged last tile."""
    rng = np.random.default_rng(1000 + hw + n)
    blk = Depth_Wise(cin, cout, residual=residual, kernel=(3, 3), stride=(stride, stride), padding=(1, 1), groups=groups)
    sd = synth_state_dict(blk.state_dict(), 900 + groups + stride + hw)
    blk.load_state_dict(sd)
    x = rng.normal(0, 1, (n, cin, hw, hw)).astype(np.float32)
    pb = PlanBuilder(n)
    inp = pb.new_buf(hw, hw, cin)
    Depth_Wise.BLOCK_SHAPES = ()       # the two-launch fp32-MFMA form (stride-2 blocks; every block with Depth_Wise.X6 off)
    Depth_Wise.X6 = False
    try:
        y = blk.emit(pb, inp.view())
    finally:
        Depth_Wise.BLOCK_SHAPES = None
        Depth_Wise.X6 = True
    plan = CompiledPlan(pb, dev)
    names = [plan.kernel_name(i) for i in range(plan.n_ops)]
    # 28 x 28 / 56 x 56 shapes: the wave-private kernel (projection weights resident in LDS); 14 x 14: the workgroup one
    want = "dwpw_persist_kernel" if groups * cout * 4 > 64 * 1024 else "dwpw_wp_kernel"
    assert any(k.startswith(want) for k in names), names
    m_rows = n * hw * hw
    assert any(k.startswith("pws_kernel") for k in names) == (m_rows % 32 == 0), names
    t = plan.buf_tensor(inp, n)
    t.zero_()
    t[..., :cin].copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    plan.run()
    torch.cuda.synchronize()
    got = plan.buf_tensor(y, n)[..., :cout].permute(0, 3, 1, 2).cpu().numpy()
    ref = mobilefacenet_ref._depth_wise({k: torch.as_tensor(v) for k, v in sd.items()}, "", torch.from_numpy(x),
                                        stride, residual).numpy()
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("cin,hw,residual,n", [
    (128, 14, True, 70),    # whole-image tiles
    (128, 14, False, 3),    # no shortcut, fewer tiles than CUs
    (64, 28, True, 37),     # four 7-row bands per image, halo rows recomputed, zero rows at the image border
    (64, 28, False, 2),
    (128, 7, True, 65),     # three images per tile; 65 = 21 * 3 + 2: the last tile holds two images
    (128, 7, False, 4),
    (128, 14, True, 530),   # bench-like batch: more tiles than resident workgroups
])
def test_dwblock_whole_depth_wise_vs_oracle(dev, cin, hw, residual, n):
    """FP_OP_DWBLOCK (csrc/dwblock.hip): expand -> depthwise -> project [+ x] of a stride-1 Depth_Wise block in one
    kernel, the expanded tensor kept in LDS, against mobilefacenet_ref._depth_wise (torch fp32 on the CPU;
    mobile_facenet.py:77-88).  Inputs have |x| ~ 1 and the output is compared to 1e-5 of its scale and, element by
    element, to 2e-5 absolute + 1e-5 relative."""
    rng = np.random.default_rng(3000 + hw + n)
    blk = Depth_Wise(cin, cin, residual=residual, kernel=(3, 3), stride=(1, 1), padding=(1, 1), groups=2 * cin)
    sd = synth_state_dict(blk.state_dict(), 1200 + cin + hw)
    blk.load_state_dict(sd)
    x = rng.normal(0, 1, (n, cin, hw, hw)).astype(np.float32)
    pb = PlanBuilder(n)
    inp = pb.new_buf(hw, hw, cin)
    Depth_Wise.BLOCK_SHAPES = (hw,)
    Depth_Wise.X6 = False
    try:
        y = blk.emit(pb, inp.view())
    finally:
        Depth_Wise.BLOCK_SHAPES = None
        Depth_Wise.X6 = True
    plan = CompiledPlan(pb, dev)
    assert plan.n_ops == 1 and plan.kernel_name(0).startswith("dwblock_kernel"), [plan.kernel_name(i) for i in range(plan.n_ops)]
    t = plan.buf_tensor(inp, n)
    t.copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    out_t = plan.buf_tensor(y, n)
    out_t.fill_(float("nan"))                   # every output element must be written
    plan.run()
    torch.cuda.synchronize()
    got = out_t.permute(0, 3, 1, 2).cpu().numpy()
    ref = mobilefacenet_ref._depth_wise({k: torch.as_tensor(v) for k, v in sd.items()}, "", torch.from_numpy(x),
                                        1, residual).numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert rel_err(got, ref) < 1e-5
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(t.permute(0, 3, 1, 2).cpu().numpy(), x)   # the input is only read


@pytest.mark.parametrize("cin,hw,residual,n", [
    (128, 14, True, 70),    # four 7x7 tiles per image (8 x 8 expand pixels each)
    (128, 14, False, 3),    # no shortcut, fewer tiles than CUs
    (64, 28, True, 37),     # four bands, interior ones with both halo rows; D / P in two row chunks
    (64, 28, False, 2),
    (128, 14, True, 530),   # bench-like batch
    (64, 28, True, 530),
    (128, 7, True, 65),     # 7x7 map: one tile per image, zero border all around
    (128, 7, False, 530),
])
def test_dwblock_x6_split_mfma_vs_oracle(dev, cin, hw, residual, n):
    """FP_OP_DWBLOCK + FP_OPF_SPLIT3 (csrc/dwblockx6.hip): the whole Depth_Wise block on the bf16 matrix cores, fp32
    operands split exactly into three bf16 pieces, six products, fp32 accumulation (csrc/split.h) -- against
    mobilefacenet_ref._depth_wise (torch fp32 on the CPU; mobile_facenet.py:77-88) to the SAME bounds as the fp32-MFMA
    kernel's test above: 1e-5 of the output scale, and element by element 2e-5 absolute + 1e-5 relative."""
    rng = np.random.default_rng(3000 + hw + n)
    blk = Depth_Wise(cin, cin, residual=residual, kernel=(3, 3), stride=(1, 1), padding=(1, 1), groups=2 * cin)
    sd = synth_state_dict(blk.state_dict(), 1200 + cin + hw)
    blk.load_state_dict(sd)
    x = rng.normal(0, 1, (n, cin, hw, hw)).astype(np.float32)
    pb = PlanBuilder(n)
    inp = pb.new_buf(hw, hw, cin)
    assert Depth_Wise.X6
    y = blk.emit(pb, inp.view())
    plan = CompiledPlan(pb, dev)
    assert plan.n_ops == 1 and plan.kernel_name(0).startswith("dwblock_x6"), [plan.kernel_name(i) for i in range(plan.n_ops)]
    t = plan.buf_tensor(inp, n)
    t.copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    out_t = plan.buf_tensor(y, n)
    out_t.fill_(float("nan"))                   # every output element must be written
    plan.run()
    torch.cuda.synchronize()
    got = out_t.permute(0, 3, 1, 2).cpu().numpy()
    ref = mobilefacenet_ref._depth_wise({k: torch.as_tensor(v) for k, v in sd.items()}, "", torch.from_numpy(x),
                                        1, residual).numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert rel_err(got, ref) < 1e-5
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(t.permute(0, 3, 1, 2).cpu().numpy(), x)   # the input is only read
    # and against fp64: not less accurate than the fp32 oracle itself
    sd64 = {k: torch.as_tensor(v).double() for k, v in sd.items()}
    ref64 = mobilefacenet_ref._depth_wise(sd64, "", torch.from_numpy(x).double(), 1, residual).numpy()
    assert np.abs(got - ref64).max() <= 2.0 * np.abs(ref - ref64).max() + 1e-6


@pytest.mark.parametrize("cin,cout,groups,hw,n", [
    (64, 128, 256, 28, 3),      # conv_34: four output bands (4, 4, 4, 2 rows)
    (64, 128, 256, 28, 70),
    (128, 128, 512, 14, 5),     # conv_45: two bands (4, 3 rows), two row parts per depthwise strip
    (128, 128, 512, 14, 530),
    (64, 128, 256, 28, 530),
    (64, 64, 128, 56, 3),       # conv_23: fourteen bands of two output rows, two column passes in the depthwise phase
    (64, 64, 128, 56, 130),
])
def test_dwblock_x6_stride2_vs_oracle(dev, cin, cout, groups, hw, n):
    """The stride-2 Depth_Wise blocks (conv_34, conv_45; mobile_facenet.py:118,123) as ONE split-MFMA kernel
    (dwblock_x6d_kernel) against mobilefacenet_ref._depth_wise, same bounds as the stride-1 form, plus the fp64 check."""
    rng = np.random.default_rng(4000 + hw + n)
    blk = Depth_Wise(cin, cout, residual=False, kernel=(3, 3), stride=(2, 2), padding=(1, 1), groups=groups)
    sd = synth_state_dict(blk.state_dict(), 1300 + cin + hw)
    blk.load_state_dict(sd)
    x = rng.normal(0, 1, (n, cin, hw, hw)).astype(np.float32)
    pb = PlanBuilder(n)
    inp = pb.new_buf(hw, hw, cin)
    y = blk.emit(pb, inp.view())
    plan = CompiledPlan(pb, dev)
    assert plan.n_ops == 1 and plan.kernel_name(0).startswith("dwblock_x6d_kernel"), [plan.kernel_name(i) for i in range(plan.n_ops)]
    t = plan.buf_tensor(inp, n)
    t.copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    out_t = plan.buf_tensor(y, n)
    out_t.fill_(float("nan"))
    plan.run()
    torch.cuda.synchronize()
    got = out_t.permute(0, 3, 1, 2).cpu().numpy()
    ref = mobilefacenet_ref._depth_wise({k: torch.as_tensor(v) for k, v in sd.items()}, "", torch.from_numpy(x), 2, False).numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert rel_err(got, ref) < 1e-5
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-5)
    sd64 = {k: torch.as_tensor(v).double() for k, v in sd.items()}
    ref64 = mobilefacenet_ref._depth_wise(sd64, "", torch.from_numpy(x).double(), 2, False).numpy()
    assert np.abs(got - ref64).max() <= 2.0 * np.abs(ref - ref64).max() + 1e-6


@pytest.mark.parametrize("hw,cu,cs,cout,n", [
    ((40, 40), 128, 256, 128, 3),     # YOLOv5n-face head, first Concat: 128 upsampled + 256 skip channels -> merged cv1 | cv2
    ((80, 80), 128, 128, 128, 2),     # second Concat (the 128-column chunk: small tiles forced)
    ((20, 20), 64, 64, 64, 5),        # yolov5n-0.5 widths: one 64-column chunk
    ((12, 10), 32, 96, 48, 7),        # non-square map, rows not a multiple of the tile, three 16-column tiles
    ((40, 40), 128, 128, 256, 2),     # two column chunks per row tile
    ((40, 40), 184, 184, 184, 2),     # yolov5s' widths: convx6_kernel, the boundary between the two tensors inside a 32-channel slab
    ((80, 80), 96, 96, 96, 2),        # ... and on a slab boundary
    ((20, 22), 40, 24, 96, 3),        # 64 input channels in two slabs, the tensor boundary at channel 40
])
def test_pwx6_upsample_folded_into_conv_is_bit_exact(dev, hw, cu, cs, cout, n):
    """FP_OPF_IN_UP2 (csrc/pwx6.hip pwx6_kernel<.., true>): a pointwise conv whose input is cat(upsample2x(u), v) reading
    the first channels from the half-size map -- nn.Upsample(None, 2, "nearest") + Concat of YOLOv5-face's head
    (y5/models/yolo.py:177-198, common.py:235-242) as operand addressing -- against the same plan with the upsampled
    slice materialised by upsample2x_kernel (bit-identical: same values in the same order) and against torch
    (interpolate + cat + conv2d + SiLU on the CPU).  The slice that is never written holds NaN in the folded plan."""
    H, W = hw
    rng = np.random.default_rng(H * 3 + cu + cout)
    u = rng.normal(0, 1, (n, cu, H // 2, W // 2)).astype(np.float32)
    v = rng.normal(0, 1, (n, cs, H, W)).astype(np.float32)
    w = rng.normal(0, (2.0 / (cu + cs)) ** 0.5, (cout, cu + cs, 1, 1)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    bias = rng.normal(0, 0.2, cout).astype(np.float32)
    got = {}
    for fold in (True, False):
        old = PlanBuilder.UP2_FOLD
        PlanBuilder.UP2_FOLD = fold
        try:
            pb = PlanBuilder(n)
            ub = pb.new_buf(H // 2, W // 2, cu)
            cb = pb.new_buf(H, W, cu + cs)
            ob = pb.new_buf(H, W, cout)
            x = cb.view()
            x.up = ub.view()
            pb.conv(x, w, ob.view(), scale=scale, bias=bias, act=L.ACT_SILU)
            assert x.up is None or fold
        finally:
            PlanBuilder.UP2_FOLD = old
        plan = CompiledPlan(pb, dev)
        names = [plan.kernel_name(i) for i in range(plan.n_ops)]
        pw = cout in (48, 64) or cout % 128 == 0          # pwx6_kernel's widths (and cu + cs a multiple of 32); else convx6_kernel
        if fold:
            assert len(names) == 1 and plan.ops[0].flags & L.OPF_IN_UP2, names
            assert names[0].startswith("pwx6_kernel") and names[0].endswith("true>") if pw else names[0].startswith("convx6_kernel"), names
        else:
            assert names[0] == "upsample2x_kernel" and not plan.ops[1].flags & L.OPF_IN_UP2, names
            assert names[1].startswith("pwx6_kernel") and names[1].endswith("false>") if pw else names[1].startswith("convx6_kernel"), names
        plan.buf_tensor(ub, n).copy_(torch.from_numpy(u).to(dev).permute(0, 2, 3, 1))
        ct = plan.buf_tensor(cb, n)
        ct[..., :cu].fill_(float("nan"))
        ct[..., cu:].copy_(torch.from_numpy(v).to(dev).permute(0, 2, 3, 1))
        ot = plan.buf_tensor(ob, n)
        ot.fill_(float("nan"))
        plan.run()
        torch.cuda.synchronize()
        got[fold] = ot.permute(0, 3, 1, 2).cpu().numpy()
        if fold:
            assert bool(torch.isnan(ct[..., :cu]).all())          # the folded plan never wrote the slice
    assert np.isfinite(got[True]).all()
    np.testing.assert_array_equal(got[True], got[False])
    xs = torch.cat([F.interpolate(torch.from_numpy(u), scale_factor=2, mode="nearest"), torch.from_numpy(v)], 1)
    r = F.conv2d(xs, torch.from_numpy(w)) * torch.from_numpy(scale).view(1, -1, 1, 1) + torch.from_numpy(bias).view(1, -1, 1, 1)
    r = (r * torch.sigmoid(r)).numpy()
    assert rel_err(got[True], r) < 1e-5


@pytest.mark.parametrize("k,n,act,res_mode,shape,in_slice", [
    (128, 128, "silu", "none", (3, 20, 20), 0),        # ragged row tile (1200 rows), one chunk
    (256, 128, "silu", "none", (2, 40, 40), 128),      # input = channel slice [128, 384) of a 384-channel concat buffer
    (384, 128, "silu", "none", (1, 17, 23), 0),        # 12 K slabs, rows not a multiple of 16
    (128, 48, "none", "none", (4, 20, 20), 0),         # Detect head: three 16-column tiles
    (256, 64, "relu", "after", (2, 16, 16), 0),        # residual added after the activation
    (128, 64, "none", "before", (2, 16, 16), 0),       # ... before it
    (256, 256, "silu", "shuffle", (2, 20, 20), 0),     # ShuffleV2 tail: out[2c] = res[c], out[2c + 1] = conv[c]; two chunks
    (128, 512, "prelu", "none", (70, 7, 7), 0),        # Mobile-FaceNet conv_6_sep: four column chunks
    (512, 512, "none", "none", (530, 1, 1), 0),        # the embedding Linear as a 1x1 conv: 16 slabs, M = 530 rows
])
def test_pwx6_pointwise_conv_vs_torch(dev, k, n, act, res_mode, shape, in_slice):
    """csrc/pwx6.hip (pointwise conv on the bf16x6 split MFMA, FP_OP_CONV + FP_OPF_SPLIT3) against torch's fp32 conv2d /
    epilogue arithmetic on the CPU, over the epilogue forms and view shapes YOLOv5-face and Mobile-FaceNet use; bounds as
    for the fp32-MFMA kernels (1e-5 of the output scale) + the fp64 check of the split arithmetic."""
    N, H, W = shape
    rng = np.random.default_rng(k * 7 + n)
    cin_phys = k + in_slice + (64 if in_slice else 0)
    x = rng.normal(0, 1, (N, cin_phys, H, W)).astype(np.float32)
    w = rng.normal(0, (2.0 / k) ** 0.5, (n, k, 1, 1)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, n).astype(np.float32)
    bias = rng.normal(0, 0.2, n).astype(np.float32)
    slope = rng.uniform(0.1, 0.3, n).astype(np.float32)
    r = rng.normal(0, 1, (N, n, H, W)).astype(np.float32)
    pb = PlanBuilder(N)
    xb = pb.new_buf(H, W, cin_phys)
    ob = pb.new_buf(H, W, 2 * n if res_mode == "shuffle" else n)
    rb = pb.new_buf(H, W, n)
    modes = {"none": L.RES_NONE, "after": L.RES_ADD_AFTER_ACT, "before": L.RES_ADD_BEFORE_ACT, "shuffle": L.RES_SHUFFLE2}
    acts = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "prelu": L.ACT_PRELU, "silu": L.ACT_SILU}
    pb.conv(xb.view(in_slice, k), w, ob.view(0, n), scale=scale, bias=bias, slope=slope if act == "prelu" else None,
            act=acts[act], res=rb.view() if res_mode != "none" else None, res_mode=modes[res_mode])
    plan = CompiledPlan(pb, dev)
    assert plan.kernel_name(0).startswith("pwx6_kernel"), plan.kernel_name(0)
    plan.buf_tensor(xb, N).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    plan.buf_tensor(rb, N).copy_(torch.from_numpy(r).to(dev).permute(0, 2, 3, 1))
    out_t = plan.buf_tensor(ob, N)
    out_t.fill_(float("nan"))
    plan.run()
    torch.cuda.synchronize()
    got = out_t.permute(0, 3, 1, 2).cpu().numpy()
    assert np.isfinite(got).all()

    def ref(dt):
        xs = torch.from_numpy(x[:, in_slice:in_slice + k]).to(dt)
        v = F.conv2d(xs, torch.from_numpy(w).to(dt)) * torch.from_numpy(scale).to(dt).view(1, -1, 1, 1) + \
            torch.from_numpy(bias).to(dt).view(1, -1, 1, 1)
        rt = torch.from_numpy(r).to(dt)
        if res_mode == "before":
            v = v + rt
        if act == "relu":
            v = torch.relu(v)
        elif act == "prelu":
            v = torch.where(v > 0, v, v * torch.from_numpy(slope).to(dt).view(1, -1, 1, 1))
        elif act == "silu":
            v = v * torch.sigmoid(v)
        if res_mode == "after":
            v = v + rt
        if res_mode == "shuffle":
            v = torch.stack([rt, v], dim=2).reshape(N, 2 * n, H, W)
        return v.numpy()
    want, want64 = ref(torch.float32), ref(torch.float64)
    assert rel_err(got, want) < 1e-5
    tol = 2e-5 if act == "silu" else 1e-5       # fp_silu: hardware exp2 / rcp (~1e-6 relative, DESIGN numerics)
    np.testing.assert_allclose(got, want, rtol=tol, atol=2e-5)
    # fp64 check: not less accurate than the fp32 reference itself (+ the hardware exp2 / rcp of fp_silu, ~1e-6 of the value)
    assert np.abs(got - want64).max() <= 2.0 * np.abs(want - want64).max() + (2e-6 if act == "silu" else 2e-7) * np.abs(want64).max()


@pytest.mark.parametrize("cin,cout,ks,stride,act,res_mode,shape", [
    (64, 64, 3, 1, "silu", "after", (2, 40, 40)),      # Bottleneck.cv2 with its shortcut (yolov5n C3)
    (128, 128, 3, 2, "silu", "none", (2, 40, 40)),     # downsampling Conv: two chunks of four column tiles
    (92, 92, 3, 1, "silu", "after", (2, 20, 24)),      # yolov5s width: Cin masked to 96, Cout padded to six tiles
    (184, 360, 3, 2, "silu", "none", (1, 21, 19)),     # odd map, 23 column tiles padded to 24, 6 channel slabs of which the last is partial
    (48, 48, 3, 1, "relu", "none", (3, 21, 21)),       # a narrow 3x3 (K < 128: only on maps of >= 400 pixels)
    (184, 184, 1, 1, "silu", "none", (2, 20, 20)),     # pointwise with a width pwx6_kernel does not take
    (720, 360, 1, 1, "silu", "none", (1, 20, 20)),     # 23 K slabs (the last one half full)
    (360, 48, 1, 1, "none", "none", (2, 10, 10)),      # Detect head of yolov5s
    (16, 32, 3, 2, "silu", "none", (2, 48, 40)),       # yolov5n stem_2b: K flattened over (tap, channel): 144 -> five slabs, two column tiles
    (24, 48, 3, 1, "relu", "none", (2, 21, 23)),       # 24 channels: a slab holds one tap and a third of the next
    (8, 32, 3, 1, "none", "after", (1, 24, 20)),       # 8 channels: four taps per slab, the ninth alone in the third
])
def test_convx6_general_conv_vs_torch(dev, cin, cout, ks, stride, act, res_mode, shape):
    """convx6_kernel (csrc/pwx6.hip: 3x3 pad-1 convs and odd-width pointwise convs on the bf16x6 split MFMA) against torch's
    fp32 conv2d on the CPU: zero padding at the borders, stride 2, Cin / Cout that are not multiples of 32 / 16."""
    N, H, W = shape
    pad = 1 if ks == 3 else 0
    OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    rng = np.random.default_rng(cin * 3 + cout + ks)
    x = rng.normal(0, 1, (N, cin, H, W)).astype(np.float32)
    w = rng.normal(0, (2.0 / (cin * ks * ks)) ** 0.5, (cout, cin, ks, ks)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    bias = rng.normal(0, 0.2, cout).astype(np.float32)
    r = rng.normal(0, 1, (N, cout, OH, OW)).astype(np.float32)
    pb = PlanBuilder(N)
    xb, ob, rb = pb.new_buf(H, W, cin), pb.new_buf(OH, OW, cout), pb.new_buf(OH, OW, cout)
    modes = {"none": L.RES_NONE, "after": L.RES_ADD_AFTER_ACT}
    acts = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "silu": L.ACT_SILU}
    pb.conv(xb.view(), w, ob.view(), stride=stride, pad=(pad, pad), scale=scale, bias=bias, act=acts[act],
            res=rb.view() if res_mode != "none" else None, res_mode=modes[res_mode])
    plan = CompiledPlan(pb, dev)
    assert plan.kernel_name(0).startswith("convx6_kernel"), plan.kernel_name(0)
    plan.buf_tensor(xb, N).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    plan.buf_tensor(rb, N).copy_(torch.from_numpy(r).to(dev).permute(0, 2, 3, 1))
    out_t = plan.buf_tensor(ob, N)
    out_t.fill_(float("nan"))
    plan.run()
    torch.cuda.synchronize()
    got = out_t.permute(0, 3, 1, 2).cpu().numpy()
    assert np.isfinite(got).all()

    def ref(dt):
        v = F.conv2d(torch.from_numpy(x).to(dt), torch.from_numpy(w).to(dt), stride=stride, padding=pad) * \
            torch.from_numpy(scale).to(dt).view(1, -1, 1, 1) + torch.from_numpy(bias).to(dt).view(1, -1, 1, 1)
        if act == "relu":
            v = torch.relu(v)
        elif act == "silu":
            v = v * torch.sigmoid(v)
        if res_mode == "after":
            v = v + torch.from_numpy(r).to(dt)
        return v.numpy()
    want, want64 = ref(torch.float32), ref(torch.float64)
    assert rel_err(got, want) < 1e-5
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    # fp64 check: not less accurate than the fp32 reference itself (+ the hardware exp2 / rcp of fp_silu, ~1e-6 of the value)
    assert np.abs(got - want64).max() <= 2.0 * np.abs(want - want64).max() + (2e-6 if act == "silu" else 2e-7) * np.abs(want64).max()


@pytest.mark.parametrize("n", [3, 70, 530])
def test_dwblock_x6_conv2_dw_in_front_vs_oracle(dev, n):
    """FP_OPF_IN_DW: Mobile-FaceNet's conv2_dw (depthwise 3x3 + BN + PReLU, mobile_facenet.py:107,141) computed in the
    prologue of conv_23's split-MFMA kernel, against the oracle's two blocks in sequence; image borders (zero padding of
    BOTH depthwise convs), first / interior / last bands."""
    from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import Conv_block
    rng = np.random.default_rng(5000 + n)
    c2 = Conv_block(64, 64, kernel=(3, 3), stride=(1, 1), padding=(1, 1), groups=64)
    c2.load_state_dict(synth_state_dict(c2.state_dict(), 1400))
    blk = Depth_Wise(64, 64, residual=False, kernel=(3, 3), stride=(2, 2), padding=(1, 1), groups=128)
    sd = synth_state_dict(blk.state_dict(), 1401)
    blk.load_state_dict(sd)
    x = rng.normal(0, 1, (n, 64, 56, 56)).astype(np.float32)
    pb = PlanBuilder(n)
    inp = pb.new_buf(56, 56, 64)
    y = blk.emit(pb, inp.view(), in_dw=c2)
    plan = CompiledPlan(pb, dev)
    assert plan.n_ops == 1 and plan.kernel_name(0) == "dwblock_x6d_kernel<64, 128, 64, 56, true>", plan.kernel_name(0)
    t = plan.buf_tensor(inp, n)
    t.copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    out_t = plan.buf_tensor(y, n)
    out_t.fill_(float("nan"))
    plan.run()
    torch.cuda.synchronize()
    got = out_t.permute(0, 3, 1, 2).cpu().numpy()

    def ref(dt):
        s2 = {k: torch.as_tensor(v).to(dt) if torch.as_tensor(v).is_floating_point() else torch.as_tensor(v) for k, v in c2.state_dict().items()}
        s3 = {k: torch.as_tensor(v).to(dt) if torch.as_tensor(v).is_floating_point() else torch.as_tensor(v) for k, v in sd.items()}
        h = mobilefacenet_ref._conv_block(s2, "", torch.from_numpy(x).to(dt), 1, 1, 64)
        return mobilefacenet_ref._depth_wise(s3, "", h, 2, False).numpy()
    want, want64 = ref(torch.float32), ref(torch.float64)
    assert got.shape == want.shape and np.isfinite(got).all()
    assert rel_err(got, want) < 1e-5
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)
    assert np.abs(got - want64).max() <= 2.0 * np.abs(want - want64).max() + 2e-7 * np.abs(want64).max()
    np.testing.assert_array_equal(t.permute(0, 3, 1, 2).cpu().numpy(), x)


@pytest.mark.parametrize("scale", [1e-4, 1.0, 3e3])
def test_split_kernels_keep_fp32_dynamic_range(dev, scale):
    """The three bf16 pieces carry fp32's exponent range, so the split kernels must be scale-invariant like an fp32 conv:
    a pointwise conv (pwx6_kernel) on inputs of magnitude `scale` with a few exact powers of two, signed zeros and values
    16 orders of magnitude apart in one row, against torch's fp32 result RELATIVE to the output scale (1e-5) and against
    fp64 (not above twice the fp32 error)."""
    rng = np.random.default_rng(17)
    N, H, W, k, n = 2, 20, 20, 128, 128
    x = (rng.normal(0, 1, (N, k, H, W)) * scale).astype(np.float32)
    x[:, 0] = scale * 2.0 ** rng.integers(-20, 20, (N, H, W))          # exact powers of two
    x[:, 1] = np.where(rng.random((N, H, W)) < 0.5, 0.0, -0.0)        # signed zeros
    x[:, 2] = scale * 1e-16 * rng.normal(0, 1, (N, H, W))             # far below the row's scale
    w = rng.normal(0, (2.0 / k) ** 0.5, (n, k, 1, 1)).astype(np.float32)
    pb = PlanBuilder(N)
    xb, ob = pb.new_buf(H, W, k), pb.new_buf(H, W, n)
    pb.conv(xb.view(), w, ob.view())
    plan = CompiledPlan(pb, dev)
    assert plan.kernel_name(0).startswith("pwx6_kernel")
    plan.buf_tensor(xb, N).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    plan.run()
    torch.cuda.synchronize()
    got = plan.buf_tensor(ob, N).permute(0, 3, 1, 2).cpu().numpy()
    want = F.conv2d(torch.from_numpy(x), torch.from_numpy(w)).numpy()
    want64 = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double()).numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, want) < 1e-5
    assert np.abs(got - want64).max() <= 2.0 * np.abs(want - want64).max() + 2e-7 * np.abs(want64).max()


def _adversarial_f32(rng, shape, kind):
    """Operands that make the split's dropped terms as large and as one-sided as they get.
    "ones": every mantissa all ones (0x..7fffff), all positive, magnitudes within a factor of 4 -- the pieces m and l sit at
    their bounds and every dropped product has the sign of the product it belongs to under a truncation split;
    "ties": mantissas on / next to the rounding ties of both cuts, all positive;  "positive": uniform in [0.5, 1)."""
    n = int(np.prod(shape))
    if kind == "positive":
        return rng.uniform(0.5, 1.0, shape).astype(np.float32)
    if kind == "ones":
        mant = np.full(n, 0x7FFFFF, np.uint32)
    else:
        mant = rng.choice(np.array([0x007FFF, 0x008000, 0x008001, 0x00807F, 0x0080FF, 0x7F7F7F, 0x7F8080], np.uint32), n)
    expo = rng.integers(125, 127, n).astype(np.uint32)              # [0.25, 1)
    return ((expo << np.uint32(23)) | mant).view(np.float32).reshape(shape)


def _fmaf_chain(a, b):
    """a (M, K) @ b (K, N) as the k-ordered fp32 fmaf chain the fp32-MFMA kernels compute (acc = fl32(acc + a_k * b_k), the
    product exact): the arithmetic the split kernels replace.  fp64 holds acc + a*b exactly enough (24 + 48 bits)."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    for k in range(a.shape[1]):
        acc = (acc.astype(np.float64) + a64[:, k, None] * b64[None, k, :]).astype(np.float32)
    return acc


@pytest.mark.parametrize("kind", ["ones", "ties", "positive"])
@pytest.mark.parametrize("k", [512, 1152])
def test_split_gemms_adversarial_vs_fp64(dev, k, kind):
    """the bf16x6 kernels on operands chosen against them -- all-ones mantissas, rounding ties, all-positive
    rows (no cancellation: every dropped term and every rounding adds up), long K -- with fp64 as the truth and NO absolute
    slack: errors are relative to sum_k |a_k b_k| and bounded in fp32 rounding units u = 2^-24.
      * against the arithmetic the split replaces (the k-ordered fp32 fmaf chain of the fp32-MFMA kernels): max and rms error
        not above twice the chain's -- or one rounding unit of the split, 2^-23, where the chain happens to be exact below
        that ("ones": every product is the same number and the chain's partial sums are representable; the split kernel
        then shows exactly what it loses: the hm + mh correction of a slab, 2^-24 of the slab, is below half an ulp of the
        running sum);
      * against the scheme's a-priori bound (split.h): dropped terms 2u + six fp32 additions per 32-k slab of half an ulp
        each = (2 + 3 K / 32) u, which is below the fmaf chain's own a-priori bound K u;
      * no drift: |mean signed error| <= 2^-23 (a truncation split drifts by -K * 2^-22 on these rows).
    torch's CPU conv (blocked partial sums) is reported in the assertion messages; DESIGN section 4 holds the measured table."""
    rng = np.random.default_rng(k + len(kind))
    N, H, W, n = 2, 16, 16, 128
    x = _adversarial_f32(rng, (N, k, H, W), kind)
    w = _adversarial_f32(rng, (n, k, 1, 1), kind)
    pb = PlanBuilder(N)
    xb, ob = pb.new_buf(H, W, k), pb.new_buf(H, W, n)
    pb.conv(xb.view(), w, ob.view())
    plan = CompiledPlan(pb, dev)
    assert plan.kernel_name(0).startswith("pwx6_kernel"), plan.kernel_name(0)
    plan.buf_tensor(xb, N).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    out_t = plan.buf_tensor(ob, N)
    out_t.fill_(float("nan"))
    plan.run()
    torch.cuda.synchronize()
    got = out_t.cpu().numpy().reshape(-1, n)                       # (pixels, cout), NHWC
    a = np.ascontiguousarray(x.transpose(0, 2, 3, 1)).reshape(-1, k)
    b = np.ascontiguousarray(w.reshape(n, k).T)
    exact = a.astype(np.float64) @ b.astype(np.float64)            # all terms positive: exact == sum |a_k b_k|
    torch32 = F.conv2d(torch.from_numpy(x), torch.from_numpy(w)).permute(0, 2, 3, 1).reshape(-1, n).numpy()
    u = 2.0 ** -24
    r6, rc, rt = (np.abs(v - exact) / exact / u for v in (got, _fmaf_chain(a, b), torch32))
    msg = {"x6 max/rms": (r6.max(), np.sqrt((r6 ** 2).mean())), "chain": (rc.max(), np.sqrt((rc ** 2).mean())),
           "torch": (rt.max(), np.sqrt((rt ** 2).mean()))}
    assert np.isfinite(got).all()
    assert r6.max() <= max(2.0 * rc.max(), 2.0) * (1 + 1e-6), msg
    assert np.sqrt((r6 ** 2).mean()) <= max(2.0 * np.sqrt((rc ** 2).mean()), 2.0) * (1 + 1e-6), msg
    assert r6.max() <= 2.0 + 3.0 * k / 32 < k, msg
    bias = ((got - exact) / exact).mean() / u
    assert abs(bias) <= 2.0 * (1 + 1e-6), (bias, msg)

    # cosine_x6_kernel on rows of the same kind (D = k): scores near 1, nothing cancels
    M, Nr = 300, 200
    G, R = _adversarial_f32(rng, (M, k), kind), _adversarial_f32(rng, (Nr, k), kind)
    best6 = S.cosine_filter(torch.from_numpy(G).to(dev), torch.from_numpy(R).to(dev), 0.05, x6=True)[0].cpu().numpy()
    best32 = S.cosine_filter(torch.from_numpy(G).to(dev), torch.from_numpy(R).to(dev), 0.05, x6=False)[0].cpu().numpy()
    G64, R64 = G.astype(np.float64), R.astype(np.float64)
    S64 = ((G64 / np.linalg.norm(G64, axis=1, keepdims=True)) @ (R64 / np.linalg.norm(R64, axis=1, keepdims=True)).T).max(1)
    Gn, Rn = G / np.linalg.norm(G, axis=1, keepdims=True), R / np.linalg.norm(R, axis=1, keepdims=True)
    cpu32 = (Gn @ Rn.T).max(1)                                       # the oracle's arithmetic (numpy fp32)
    c6, c32, ccpu = (np.abs(v - S64).max() / u for v in (best6, best32, cpu32))   # scores in [0, 1]: sum |a_k b_k| = 1 after the norms
    assert c6 <= max(2.0 * c32, 4.0) and c6 <= max(2.0 * ccpu, 4.0), (c6, c32, ccpu)   # (two more roundings: the inverse norms)


@pytest.mark.parametrize("cin,hw", [(128, 14), (64, 28), (128, 7)])
def test_dwblock_x6_adversarial_vs_fp64(dev, cin, hw):
    """The whole-Depth_Wise split kernel (dwblock_x6_kernel / dwblock_x6q_kernel) on all-positive, all-ones-mantissa weights
    and inputs (both GEMMs, K = cin and 2 cin, see no cancellation; BN is the identity, PReLU never fires), against the fp64
    oracle with no absolute slack: not above twice the fp32 oracle's own error."""
    rng = np.random.default_rng(77 + hw)
    n = 5
    blk = Depth_Wise(cin, cin, residual=True, kernel=(3, 3), stride=(1, 1), padding=(1, 1), groups=2 * cin)
    sd = {k_: torch.as_tensor(v).clone() for k_, v in synth_state_dict(blk.state_dict(), 1).items()}
    for k_, v in sd.items():
        if k_.endswith("conv.weight"):
            fan = v[0].numel()
            sd[k_] = torch.from_numpy(_adversarial_f32(rng, tuple(v.shape), "ones") * np.float32(2.0 / fan))
        elif k_.endswith("bn.weight") or k_.endswith("running_var"):
            sd[k_] = torch.ones_like(v)
        elif k_.endswith("bn.bias") or k_.endswith("running_mean"):
            sd[k_] = torch.zeros_like(v)
    blk.load_state_dict(sd)
    x = _adversarial_f32(rng, (n, cin, hw, hw), "ones")
    pb = PlanBuilder(n)
    inp = pb.new_buf(hw, hw, cin)
    y = blk.emit(pb, inp.view())
    plan = CompiledPlan(pb, dev)
    assert plan.n_ops == 1 and plan.kernel_name(0).startswith("dwblock_x6"), plan.kernel_name(0)
    plan.buf_tensor(inp, n).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    out_t = plan.buf_tensor(y, n)
    out_t.fill_(float("nan"))
    plan.run()
    torch.cuda.synchronize()
    got = out_t.permute(0, 3, 1, 2).cpu().numpy()
    ref = mobilefacenet_ref._depth_wise(sd, "", torch.from_numpy(x), 1, True).numpy()
    ref64 = mobilefacenet_ref._depth_wise({k_: (v.double() if v.is_floating_point() else v) for k_, v in sd.items()}, "",
                                          torch.from_numpy(x).double(), 1, True).numpy()
    assert np.isfinite(got).all() and (ref64 > 0).all()
    u = 2.0 ** -24
    e6, e32 = (np.abs(got - ref64) / ref64).max() / u, (np.abs(ref - ref64) / ref64).max() / u
    assert e6 <= max(2.0 * e32, 4.0), (e6, e32)     # two split GEMMs in sequence: 2 x 2^-23 where the fp32 oracle is exact below that


def test_mobilefacenet_forward_with_whole_block_kernels_vs_reference_golden(dev):
    """The reference's own Mobile-FaceNet output (tests/golden/mobilefacenet_forward.npz) through three plans (batch
    capacity 64, run on the golden's 4 images): (a) the default -- all twelve stride-1 blocks as the
    bf16x6 split-MFMA kernel (FP_OPF_SPLIT3), (b) Depth_Wise.X6 off and all twelve stride-1 blocks as the fp32-MFMA
    FP_OP_DWBLOCK, (c) X6 off, two-launch form.  Each within the north_star's 1e-4 of the reference (measured ~4e-7),
    2e-6 of each other."""
    g = golden("mobilefacenet_forward")
    net = MobileFaceNet(512)
    net.load_state_dict(synth_state_dict(net.state_dict(), int(g["seed"])))
    net = net.to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    n = x.shape[0]

    def run(plan):
        plan.input[:n, ..., :3].copy_(x.permute(0, 2, 3, 1))
        plan.input[:n, ..., 3:].zero_()
        plan.run(n)
        torch.cuda.synchronize()
        return plan.out[:n].cpu().numpy().copy()

    plan = net.plan_for(64)
    ops = [plan.ops[i] for i in range(plan.n_ops)]
    assert sum(1 for o in ops if o.kind == L.OP_DWBLOCK and o.flags & L.OPF_SPLIT3) == 15   # + conv_23, conv_34, conv_45
    assert plan.kernel_name(0) == "stemdw_kernel<true>" and not any(o.flags & L.OPF_IN_DW for o in ops)   # conv1 + conv2_dw: one kernel
    e_x6 = run(plan)
    MobileFaceNet.STEM_DW = False          # round 3's form: conv2_dw in the prologue of conv_23's kernel (FP_OPF_IN_DW)
    try:
        plan = net.plan_for(64)
        assert any(plan.ops[i].flags & L.OPF_IN_DW for i in range(plan.n_ops)) and not plan.kernel_name(0).startswith("stemdw_kernel")
        e_indw = run(plan)
    finally:
        MobileFaceNet.STEM_DW = True
    assert np.abs(e_indw - g["emb"]).max() < 1e-4 and np.abs(e_indw - e_x6).max() < 2e-6
    Depth_Wise.X6 = False
    try:
        Depth_Wise.BLOCK_SHAPES = (28, 14, 7)
        plan = net.plan_for(64)
        ops = [plan.ops[i] for i in range(plan.n_ops)]
        assert sum(1 for o in ops if o.kind == L.OP_DWBLOCK) == 12 and not any(o.flags & L.OPF_SPLIT3 for o in ops)
        e_blk = run(plan)
        Depth_Wise.BLOCK_SHAPES = ()
        plan = net.plan_for(n)
        assert L.OP_DWBLOCK not in [plan.ops[i].kind for i in range(plan.n_ops)]
        e_two = run(plan)
    finally:
        Depth_Wise.BLOCK_SHAPES = None
        Depth_Wise.X6 = True
    for e in (e_x6, e_blk, e_two):
        assert np.abs(e - g["emb"]).max() < 1e-4
    assert np.abs(e_blk - e_two).max() < 2e-6 and np.abs(e_x6 - e_two).max() < 2e-6


@pytest.mark.parametrize("x6", [True, False])
@pytest.mark.parametrize("n", [1, 3, 70, 530])
def test_stemdw_conv1_plus_conv2_dw_in_one_kernel_vs_oracle(dev, n, x6):
    """FP_OPF_OUT_DW (csrc/stemdw.hip): Mobile-FaceNet's conv1 (3x3 stride 2 + BN + PReLU, mobile_facenet.py:107,141) and
    conv2_dw (depthwise 3x3 + BN + PReLU, :108,142) in ONE kernel -- conv1's rows never leave LDS -- against the oracle's two
    Conv_blocks in sequence (1e-5 of the output scale, element by element 2e-5 + 1e-5 relative), against the two separate
    launches (stem kernel, then depthwise kernel: 2e-6), every output written, the image borders (zero padding of BOTH convs)
    and first / interior / last bands; batches below and above one round of workgroups.  x6: conv1 on the bf16 matrix cores
    with the exact three-way operand split (stemdw_kernel<true>, the default) / on the fp32 MFMA (PlanBuilder.X6 off)."""
    from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import Conv_block, _affine
    from face_detection_and_recognition_amd.modules.params import npy
    rng = np.random.default_rng(900 + n)
    c1 = Conv_block(3, 64, kernel=(3, 3), stride=(2, 2), padding=(1, 1))
    c2 = Conv_block(64, 64, kernel=(3, 3), stride=(1, 1), padding=(1, 1), groups=64)
    sd1, sd2 = synth_state_dict(c1.state_dict(), 1500), synth_state_dict(c2.state_dict(), 1501)
    c1.load_state_dict(sd1)
    c2.load_state_dict(sd2)
    x = rng.uniform(-1, 1, (n, 3, 112, 112)).astype(np.float32)
    x[0, :, :3, :] = 1.0                      # a bright top edge / corner: the zero padding must show in the borders
    x[-1, :, -2:, -2:] = -1.0

    def run(fused):
        pb = PlanBuilder(n)
        inp = pb.new_buf(112, 112, 3)
        if fused:
            y = pb.new_buf(56, 56, 64)
            s1, b1 = _affine(c1.bn)
            pb.conv(inp.view(), npy(c1.conv.weight), y.view(), stride=2, pad=(1, 1), scale=s1, bias=b1, slope=npy(c1.prelu.weight),
                    act=L.ACT_PRELU, out_dw=(npy(c2.conv.weight), _affine(c2.bn), npy(c2.prelu.weight)))
        else:
            y = c2.emit(pb, c1.emit(pb, inp.view()).view())
        plan = CompiledPlan(pb, dev)
        t = plan.buf_tensor(inp, n)
        t[..., :3].copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
        t[..., 3:].zero_()
        out_t = plan.buf_tensor(y, n)
        out_t.fill_(float("nan"))
        plan.run()
        torch.cuda.synchronize()
        return plan, out_t.permute(0, 3, 1, 2).cpu().numpy()
    x6_was, PlanBuilder.X6 = PlanBuilder.X6, x6
    try:
        plan, got = run(True)
    finally:
        PlanBuilder.X6 = x6_was
    assert plan.n_ops == 1 and plan.kernel_name(0) == ("stemdw_kernel<true>" if x6 else "stemdw_kernel<false>")
    assert plan.ops[0].flags & L.OPF_OUT_DW and bool(plan.ops[0].flags & L.OPF_SPLIT3) == x6
    plan2, two = run(False)
    assert plan2.n_ops == 2
    xt = torch.from_numpy(x)
    sd = {"conv1." + k: torch.as_tensor(v) for k, v in sd1.items()}
    sd.update({"conv2_dw." + k: torch.as_tensor(v) for k, v in sd2.items()})
    ref = mobilefacenet_ref._conv_block(sd, "conv2_dw.", mobilefacenet_ref._conv_block(sd, "conv1.", xt, 2, 1, 1), 1, 1, 64).numpy()
    assert got.shape == ref.shape == (n, 64, 56, 56) and np.isfinite(got).all()
    assert rel_err(got, ref) < 1e-5
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(got, two, rtol=0, atol=2e-6 * np.abs(ref).max())
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = mobilefacenet_ref._conv_block(sd64, "conv2_dw.", mobilefacenet_ref._conv_block(sd64, "conv1.", xt.double(), 2, 1, 1),
                                          1, 1, 64).numpy()
    assert np.abs(got - ref64).max() <= 2.0 * np.abs(ref - ref64).max()       # plain fp32 arithmetic: no worse than the oracle's


@pytest.mark.parametrize("ks,cout,hw,n,act", [(5, 24, 256, 8, "relu"), (3, 64, 112, 48, "prelu"), (3, 24, 90, 70, "none")])
def test_stem_conv_kernel_vs_torch(dev, ks, cout, hw, n, act):
    """csrc/stem.hip (KxK stride-2 conv on the 4-float-pixel image, LDS-staged window) against torch's fp32 conv2d on
    the CPU: BlazeFace's 5x5 stem, Mobile-FaceNet's 3x3 stem (ragged last tile: 3136 = 24.5 x 128), and an odd size."""
    rng = np.random.default_rng(ks * 100 + hw)
    x = rng.normal(0, 1, (n, 3, hw, hw)).astype(np.float32)
    w = rng.normal(0, 0.2, (cout, 3, ks, ks)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    bias = rng.normal(0, 0.3, cout).astype(np.float32)
    slope = rng.uniform(0.05, 0.4, cout).astype(np.float32)
    oh = (hw + 2 - ks) // 2 + 1 if ks == 3 else (hw + 3 - ks) // 2 + 1     # pad 1 / F.pad(1, 2, 1, 2)
    pb = PlanBuilder(n)
    inp = pb.new_buf(hw, hw, 3)
    out = pb.new_buf(oh, oh, cout)
    pb.conv(inp.view(), w, out.view(), stride=2, pad=(1, 1), scale=scale, bias=bias,
            slope=slope if act == "prelu" else None,
            act={"relu": L.ACT_RELU, "prelu": L.ACT_PRELU, "none": L.ACT_NONE}[act])
    plan = CompiledPlan(pb, dev)
    assert plan.kernel_name(0).startswith("stem_conv_kernel"), plan.kernel_name(0)
    t = plan.buf_tensor(inp, n)
    t.zero_()
    t[..., :3].copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    plan.run()
    torch.cuda.synchronize()
    got = plan.buf_tensor(out, n)[..., :cout].permute(0, 3, 1, 2).cpu().numpy()
    xt = torch.from_numpy(x)
    xt = torch.nn.functional.pad(xt, (1, 2, 1, 2)) if ks == 5 else torch.nn.functional.pad(xt, (1, 1, 1, 1))
    ref = torch.nn.functional.conv2d(xt, torch.from_numpy(w), stride=2)[:, :, :oh, :oh]
    ref = ref * torch.from_numpy(scale).view(1, -1, 1, 1) + torch.from_numpy(bias).view(1, -1, 1, 1)
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "prelu":
        ref = torch.where(ref > 0, ref, ref * torch.from_numpy(slope).view(1, -1, 1, 1))
    assert got.shape == tuple(ref.shape)
    assert rel_err(got, ref.numpy()) < 1e-5


@pytest.mark.parametrize("n,hw", [(48, 56), (2, 56), (3, 14)])
def test_dwpw_with_output_prelu_vs_torch(dev, n, hw):
    """FP_OP_DWPW with bias_off (output PReLU): depthwise Conv_block -> 1x1 Conv_block (conv2_dw -> conv_23.conv,
    mobile_facenet.py:117-118,70) in one kernel, persistent (n = 48) and per-tile (small n) variants, vs torch fp32."""
    rng = np.random.default_rng(77 + n)
    G, cout = 64, 128
    x = rng.normal(0, 1, (n, G, hw, hw)).astype(np.float32)
    dw_w = rng.normal(0, 0.3, (G, 1, 3, 3)).astype(np.float32)
    pw_w = rng.normal(0, 0.15, (cout, G, 1, 1)).astype(np.float32)
    ds, db, dsl = (rng.uniform(0.5, 1.5, G).astype(np.float32), rng.normal(0, 0.2, G).astype(np.float32),
                   rng.uniform(0.05, 0.5, G).astype(np.float32))
    ps, pbi, psl = (rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.2, cout).astype(np.float32),
                    rng.uniform(0.05, 0.5, cout).astype(np.float32))
    pb = PlanBuilder(n)
    inp = pb.new_buf(hw, hw, G)
    out = pb.new_buf(hw, hw, cout)
    pb.dwpw(inp.view(), dw_w, ds, db, dsl, pw_w, ps, pbi, out.view(), 1, out_slope=psl)
    plan = CompiledPlan(pb, dev)
    assert plan.kernel_name(0).startswith("dwpw_wp_kernel") == (n * hw * hw >= 131072), plan.kernel_name(0)
    t = plan.buf_tensor(inp, n)
    t.copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    plan.run()
    torch.cuda.synchronize()
    got = plan.buf_tensor(out, n).permute(0, 3, 1, 2).cpu().numpy()
    F = torch.nn.functional
    v = lambda a: torch.from_numpy(a).view(1, -1, 1, 1)
    y = F.conv2d(torch.from_numpy(x), torch.from_numpy(dw_w), padding=1, groups=G) * v(ds) + v(db)
    y = torch.where(y > 0, y, y * v(dsl))
    y = F.conv2d(y, torch.from_numpy(pw_w)) * v(ps) + v(pbi)
    y = torch.where(y > 0, y, y * v(psl))
    assert rel_err(got, y.numpy()) < 1e-5


def test_mobilefacenet_forward_vs_reference_golden(dev):
    g = golden("mobilefacenet_forward")
    net = MobileFaceNet(512)
    net.load_state_dict(synth_state_dict(net.state_dict(), int(g["seed"])))
    net = net.to(dev)
    e = net(torch.from_numpy(g["x"]))
    torch.cuda.synchronize()
    e = e.cpu().numpy()
    assert np.abs(e - g["emb"]).max() < 1e-4                        # north_star tolerance
    np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)


def test_plan_runs_on_a_prefix_of_its_capacity(dev):
    """CompiledPlan.run(n): a plan emitted for a capacity of 40 crops processes the first 16 / 24 of them and gives
    exactly what a plan emitted for that batch size gives (FacePipeline keeps ONE embedder plan for every face count)."""
    net = MobileFaceNet(512)
    net.load_state_dict(synth_state_dict(net.state_dict(), 300))
    net = net.to(dev)
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.uniform(-1, 1, (40, 112, 112, 3)).astype(np.float32)).to(dev)
    big = net.plan_for(40)
    big.input.zero_()
    big.input[..., :3].copy_(x)
    for n in (16, 24, 40):
        big.run(n=n)
        got = big.out[:n].clone()
        small = net.plan_for(n)
        small.input.zero_()
        small.input[..., :3].copy_(x[:n])
        small.run()
        torch.cuda.synchronize()
        assert torch.equal(got, small.out[:n])
    with pytest.raises(ValueError):
        big.run(n=41)


@pytest.mark.parametrize("n", [8, 264, 520, 528])
def test_embedder_prefix_run_equals_exact_size_plan(dev, n):
    """FacePipeline runs ONE embedder plan (capacity in steps of 256 crops) on the step's face count: kernel selection
    (streaming / persistent / per-tile / whole-block) follows the batch it runs on, not the capacity, so a prefix run
    must give the SAME embeddings as a plan built for exactly that batch -- across the dispatch thresholds (8: per-tile
    kernels; 264 / 520 / 528: persistent + streaming kernels, the 7x7 blocks as FP_OP_DWBLOCK) and with junk in the rows
    past n.  Bit-identical: same kernels on the same rows."""
    net = MobileFaceNet(512)
    net.load_state_dict(synth_state_dict(net.state_dict(), 77))
    net = net.to(dev)
    x = torch.randn((n, 112, 112, 4), device=dev)
    x[..., 3] = 0
    big = net.plan_for(768, n_run=n)
    big.input.fill_(3.0)                       # whatever an earlier, larger step left behind
    big.input[:n].copy_(x)
    big.run(n)
    torch.cuda.synchronize()
    e_big = big.out[:n].clone()
    exact = net.plan_for(n)
    assert [exact.ops[i].kind for i in range(exact.n_ops)] == [big.ops[i].kind for i in range(big.n_ops)]
    exact.input.copy_(x)
    exact.run()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(e_big.cpu().numpy(), exact.out.cpu().numpy())
    np.testing.assert_allclose(np.linalg.norm(e_big.cpu().numpy(), axis=1), 1.0, atol=1e-5)


def test_blazeface_plan_with_row_padded_buffers_runs_on_a_prefix(dev):
    """A BlazeFace plan of capacity 40 (persistent stride-2 kernel at n = 40, per-tile kernel at n = 3: both write the
    row-padded layout; the wave-private kernels and the stem follow the actual n) run on 3, then 40, then 17 images
    gives what plans emitted for those batch sizes give: per-image strides and pads do not depend on the batch."""
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(0, 256, (40, 256, 256, 3), dtype=np.uint8)).to(dev)
    net = BlazeFace(True)
    net.load_state_dict(synth_state_dict(net.state_dict(), 11, residual_gain=0.5))
    net = net.to(dev)
    big = net.plan_for(40)
    assert any(big.ops[i].flags for i in range(big.n_ops))
    lut = torch.from_numpy(image_ref.blaze_lut()).to(dev)
    for n in (3, 40, 17):
        big.input[:n, ..., :3].copy_(lut[x[:n].long()])
        big.run(n=n)
        got_r, got_c = big.r[:n].clone(), big.c[:n].clone()
        small = net.plan_for(n)
        small.input[..., :3].copy_(lut[x[:n].long()])
        small.run()
        torch.cuda.synchronize()
        assert torch.equal(got_r, small.r[:n]) and torch.equal(got_c, small.c[:n])


def test_resize_normalize_vs_oracle(dev, lib):
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (3, 72, 128, 3), dtype=np.uint8)
    f = torch.from_numpy(frames).to(dev)
    from face_detection_and_recognition_amd.modules.utils.image import letterbox_batch
    for (nw, nh) in ((32, 32), (64, 48), (200, 100), (128, 72)):
        canvas = torch.zeros((3, nh, nw, 4), device=dev)
        lut = torch.from_numpy(image_ref.blaze_lut()).to(dev)
        letterbox_batch(f, (nw, nh), lut, canvas, pad_value=125, swap_rb=True)
        torch.cuda.synchronize()
        for i in range(3):
            ref = image_ref.pad_resize_image(frames[i], (nw, nh))[..., ::-1]
            np.testing.assert_array_equal(canvas[i, ..., :3].cpu().numpy(), image_ref.blaze_lut()[ref])
            assert float(canvas[i, ..., 3].abs().max()) == 0.0


@pytest.mark.parametrize("frame_hw,canvas_hw,n_items", [((72, 128), (112, 112), 37), ((576, 1024), (112, 112), 64),
                                                         ((33, 5), (40, 24), 9), ((90, 160), (53, 37), 11),
                                                         ((120, 200), (640, 640), 3)])
def test_resize_tabled_kernel_is_bit_exact(dev, lib, frame_hw, canvas_hw, n_items):
    """fp_resize_normalize's tabled kernel (csrc/image.hip resize_normalize_rows_kernel: taps of an item computed once per
    workgroup, 8-byte window loads) against the per-pixel kernel (FP_RESIZE_PER_PIXEL=1; checked against the oracle's
    cv2 restatement in test_resize_normalize_vs_oracle) on random crop items: up- and down-scaling, source rectangles
    touching and crossing the frame border (clamped in the kernel), destination rectangles smaller than the canvas (pad
    colour around them), empty destinations, source image indices out of range.  Every byte equal."""
    import os
    fh, fw = frame_hw
    ch, cw = canvas_hw
    rng = np.random.default_rng(fh * 7 + cw)
    frames = torch.from_numpy(rng.integers(0, 256, (4, fh, fw, 3), dtype=np.uint8)).to(dev)
    items = np.zeros((n_items, 9), np.int32)
    for k in range(n_items):
        sw, sh = int(rng.integers(1, fw + 1)), int(rng.integers(1, fh + 1))
        sx, sy = int(rng.integers(-3, fw - sw + 4)), int(rng.integers(-3, fh - sh + 4))      # may stick out of the frame
        dw, dh = int(rng.integers(0, cw + 1)), int(rng.integers(0, ch + 1))
        dx, dy = int(rng.integers(0, cw - dw + 1)), int(rng.integers(0, ch - dh + 1))
        items[k] = [int(rng.integers(-1, 5)), sx, sy, sw, sh, dx, dy, dw, dh]
    items[0] = [1, 0, 0, fw, fh, 0, 0, cw, ch]              # the whole frame onto the whole canvas
    items[1] = [2, fw - 1, fh - 1, 1, 1, 0, 0, cw, ch]      # one source pixel (the last byte of a frame row in its window)
    it = torch.from_numpy(items).to(dev)
    lut = torch.from_numpy(image_ref.blaze_lut()).to(dev)
    out = {}
    for per_pixel in (False, True):
        if per_pixel:
            os.environ["FP_RESIZE_PER_PIXEL"] = "1"
        lib.fp_debug_reload_env()          # the library reads its knobs once at load; this re-reads them
        try:
            canvas = torch.full((n_items, ch, cw, 4), float("nan"), device=dev)
            L.check(lib.fp_resize_normalize(L.ptr(frames), 4, fh, fw, L.ptr(it), n_items, L.ptr(canvas), ch, cw, 4, L.ptr(lut),
                                            125, 1, L.current_stream(dev)), "fp_resize_normalize")
            torch.cuda.synchronize()
        finally:
            os.environ.pop("FP_RESIZE_PER_PIXEL", None)
            lib.fp_debug_reload_env()
        out[per_pixel] = canvas.cpu().numpy()
    assert np.isfinite(out[False]).all()
    np.testing.assert_array_equal(out[False], out[True])


def test_similarity_vs_golden_and_oracle(dev):
    """fp_l2_mean_thres / fp_l2_filter against what the reference's own get_ref_mean_vec_and_thres_from_imgs and main()
    produced (tests/golden/similarity.npz, three classes; sff/filter_faces_using_reference.py:71-100,183-197): mean to
    1e-6, threshold to 1e-6 relative, the clean / unclean decision of every image exact except where the reference's
    distance is within 1e-4 of its threshold -- and the reference images that reappear among the unfiltered ones
    (class 2, distance == threshold for the farthest) must be kept."""
    g = golden("similarity")
    for c in range(3):
        ref = torch.from_numpy(g[f"c{c}_ref"]).to(dev)
        mean, thres = S.l2_mean_thres(ref)
        np.testing.assert_allclose(mean.cpu().numpy(), g[f"c{c}_mean"], rtol=0, atol=1e-6)
        assert abs(float(thres) - float(g[f"c{c}_thres"])) < 1e-6 * float(g[f"c{c}_thres"]) + 1e-6
        E = g[f"c{c}_E"]
        dist, keep = S.l2_filter(torch.from_numpy(E).to(dev), mean, thres)
        dref = np.array([np.linalg.norm(e - g[f"c{c}_mean"]) for e in E])
        np.testing.assert_allclose(dist.cpu().numpy(), dref, rtol=0, atol=1e-4)
        sure = np.abs(dref - float(g[f"c{c}_thres"])) > 1e-4
        assert sure.sum() >= len(sure) - 1
        np.testing.assert_array_equal(keep.cpu().numpy()[sure], g[f"c{c}_keep"][sure])
        if c == 2:
            assert keep.cpu().numpy()[:6].all()
    # cosine vs the reference formula (pairwise) and the oracle
    best, arg, keepc = S.cosine_filter(torch.from_numpy(g["cos_a"]).to(dev), torch.from_numpy(g["cos_b"]).to(dev), 0.1)
    sim = 1.0 - g["cos_dist"]
    np.testing.assert_allclose(best.cpu().numpy(), sim.max(1), rtol=0, atol=1e-4)
    np.testing.assert_array_equal(arg.cpu().numpy(), sim.argmax(1))


def test_cosine_filter_ragged_and_large(dev):
    rng = np.random.default_rng(9)
    for M, Nr, D in ((1, 1, 512), (130, 257, 512), (1000, 77, 128), (4096, 3000, 512)):
        G = rng.normal(0, 1, (M, D)).astype(np.float32)
        R = rng.normal(0, 1, (Nr, D)).astype(np.float32)
        rb, ra, rk, Smat = similarity_ref.cosine_filter(G, R, 0.05)
        S64 = (G.astype(np.float64) / np.linalg.norm(G.astype(np.float64), axis=1, keepdims=True)) @ \
              (R.astype(np.float64) / np.linalg.norm(R.astype(np.float64), axis=1, keepdims=True)).T
        got = {}
        for x6 in (False, True):   # fp32-MFMA kernel / split-MFMA kernel (csrc/sim.hip cosine_x6_kernel; D = 512 and 128)
            best, arg, keep = S.cosine_filter(torch.from_numpy(G).to(dev), torch.from_numpy(R).to(dev), 0.05, x6=x6)
            got[x6] = best.cpu().numpy()
            np.testing.assert_allclose(got[x6], rb, rtol=0, atol=1e-4)
            a = arg.cpu().numpy()
            # the argmax may differ only where two scores tie within rounding
            bad = a != ra
            assert np.all(np.abs(Smat[np.arange(M), a][bad] - rb[bad]) < 1e-5)
            np.testing.assert_array_equal(keep.cpu().numpy(), got[x6] >= 0.05)
        # the split arithmetic is not less accurate than the fp32 chain (fp64 as the truth)
        e32, e6 = np.abs(got[False] - S64.max(1)).max(), np.abs(got[True] - S64.max(1)).max()
        assert e6 <= 2.0 * e32 + 1e-7, (e6, e32)


# ------------------------------------------------------------------ YOLOv5-face
def _yolo(name, dev, seed, fuse):
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import Model
    m = Model(name)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed))
    if fuse:
        m.fuse()
    return m.to(dev)


@pytest.mark.parametrize("name", ["yolov5n", "yolov5s", "yolov5n-0.5"])
@pytest.mark.parametrize("fuse", [True, False])
def test_yolo_forward_vs_reference_golden(dev, name, fuse):
    g = golden(f"{name}_forward")
    m = _yolo(name, dev, int(g["seed"]), fuse)
    z, heads = m(torch.from_numpy(g["x"]))
    torch.cuda.synchronize()
    for i, h in enumerate(heads):
        assert rel_err(h.cpu().numpy(), g[f"head{i}"]) < 1e-4
    assert rel_err(z.cpu().numpy(), g["z"]) < 1e-4
    # absolute bounds on the decoded rows (yolo.py:62-108): objectness / class confidence (columns 4, 15) within 1e-4
    # (north_star), boxes and landmarks within 1e-4 of the 128-pixel input (0.0128 px)
    zz, zr = z.cpu().numpy(), g["z"]
    assert np.abs(zz[..., [4, 15]] - zr[..., [4, 15]]).max() < 1e-4
    assert np.abs(np.delete(zz, [4, 15], axis=-1) - np.delete(zr, [4, 15], axis=-1)).max() < 1e-4 * 128


def _run_yolo_block(block, x_nchw, dev):
    """One emit()-able YOLOv5-face block on an NCHW numpy input -> NCHW numpy (all logical output channels)."""
    N, C, H, W = x_nchw.shape
    pb = PlanBuilder(N)
    inp = pb.new_buf(H, W, C)
    y = block.emit(pb, inp.view())
    plan = CompiledPlan(pb, dev)
    t = plan.buf_tensor(inp, N)
    t.zero_()
    t[..., :C].copy_(torch.from_numpy(x_nchw).to(dev).permute(0, 2, 3, 1))
    plan.run()
    torch.cuda.synchronize()
    return plan.buf_tensor(y.buf, N)[..., y.coff:y.coff + y.C].permute(0, 3, 1, 2).cpu().numpy()


_YOLO_BLOCKS = {
    "conv": lambda Y: Y.Conv(16, 32, 3, 2), "conv1x1": lambda Y: Y.Conv(24, 40, 1, 1),
    "stem": lambda Y: Y.StemBlock(3, 32, 3, 2), "shuffle_s2": lambda Y: Y.ShuffleV2Block(32, 128, 2),
    "shuffle_s1": lambda Y: Y.ShuffleV2Block(128, 128, 1), "c3": lambda Y: Y.C3(64, 64, 2),
    "c3_noshortcut": lambda Y: Y.C3(96, 64, 1, False), "spp": lambda Y: Y.SPP(128, 128, (3, 5, 7)),
}


@pytest.mark.parametrize("tag", sorted(_YOLO_BLOCKS))
def test_yolo_block_vs_reference_golden(dev, tag):
    """each YOLOv5-face block on the HIP path against the reference's own module output
    (tests/golden/yolo_blocks.npz; Conv also after fuse_conv_and_bn)."""
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    g = golden("yolo_blocks")
    blk = _YOLO_BLOCKS[tag](Y)
    blk.load_state_dict(synth_state_dict(blk.state_dict(), int(g[f"{tag}_seed"])))
    want = g[f"{tag}_y"]
    got = _run_yolo_block(blk.to(dev), g[f"{tag}_x"], dev)[:, :want.shape[1]]
    assert rel_err(got, want) < 1e-5, tag
    if tag == "conv":
        blk.fuse()
        got = _run_yolo_block(blk, g["conv_x"], dev)[:, :want.shape[1]]
        assert rel_err(got, g["conv_y_fused"]) < 1e-5
    if tag.startswith("c3"):               # cv1 / cv2 as separate convs (default: one conv writing whole concat rows)
        Y.C3.MERGE = False
        try:
            got = _run_yolo_block(blk, g[f"{tag}_x"], dev)[:, :want.shape[1]]
        finally:
            Y.C3.MERGE = True
        assert rel_err(got, want) < 1e-5, tag
    if tag.startswith("shuffle"):          # the unfused DWCONV + CONV form of the branches (default: FP_OP_DWPW)
        Y.ShuffleV2Block.FUSE = False
        try:
            got = _run_yolo_block(blk, g[f"{tag}_x"], dev)[:, :want.shape[1]]
        finally:
            Y.ShuffleV2Block.FUSE = True
        assert rel_err(got, want) < 1e-5, tag


def test_yolo_shufflev2_fused_branches_at_stage_shapes(dev):
    """Stage-3 shapes of yolov5n (128 -> 256 stride 2 with BOTH branches fused, then 256 -> 256 stride 1) on a ragged
    map: FP_OP_DWPW with the SiLU / FP_RES_SHUFFLE2 epilogue against the oracle block."""
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    from oracle import yolo_ref
    rng = np.random.default_rng(12)
    for blk, cin, stride in ((Y.ShuffleV2Block(128, 256, 2), 128, 2), (Y.ShuffleV2Block(256, 256, 1), 256, 1)):
        blk.load_state_dict(synth_state_dict(blk.state_dict(), 640 + stride))
        x = rng.normal(0, 1, (3, cin, 22, 26)).astype(np.float32)
        pb = PlanBuilder(3)
        assert sum(1 for op in (blk.emit(pb, pb.new_buf(22, 26, cin).view()), pb)[1].ops if op.kind == L.OP_DWPW) == \
            (2 if stride == 2 else 1)
        got = _run_yolo_block(blk.to(dev), x, dev)
        with torch.no_grad():
            want = yolo_ref._shuffle_block({k: v.cpu() for k, v in blk.state_dict().items()}, "", torch.from_numpy(x),
                                           stride).numpy()
        assert rel_err(got[:, :want.shape[1]], want) < 1e-5


@pytest.mark.parametrize("hw,n", [((22, 26), 3), ((64, 96), 2), ((2, 2), 1), ((160, 160), 5), ((10, 34), 4)])
def test_yolo_shufflev2_stride2_block_in_one_kernel_vs_oracle(dev, hw, n):
    """FP_OP_SHUFDOWN (csrc/shufdown.hip): YOLOv5n-face's first ShuffleV2Block (32 -> 128 channels, stride 2: both branches, cat
    and channel_shuffle in one kernel, split-MFMA 1x1 convs) against the oracle block (y5/models/common.py:127-176 restated) and
    against the four-op form, on maps whose 4 x 16 output tiles overhang (11 x 13, 1 x 1, 5 x 17), divide (32 x 48) and at the
    network's own 160 x 160."""
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    from oracle import yolo_ref
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    blk = Y.ShuffleV2Block(32, 128, 2)
    blk.load_state_dict(synth_state_dict(blk.state_dict(), 660 + hw[0]))
    x = rng.normal(0, 1, (n, 32) + hw).astype(np.float32)
    pb = PlanBuilder(n)
    blk.emit(pb, pb.new_buf(hw[0], hw[1], 32).view())
    assert [op.kind for op in pb.ops] == [L.OP_SHUFDOWN] and pb.ops[0].flags == L.OPF_SPLIT3
    got = _run_yolo_block(blk.to(dev), x, dev)
    with torch.no_grad():
        want = yolo_ref._shuffle_block({k: v.cpu() for k, v in blk.state_dict().items()}, "", torch.from_numpy(x), 2).numpy()
    assert got.shape[0] == n and got[:, :128].shape == want.shape
    assert rel_err(got[:, :128], want) < 1e-5
    Y.ShuffleV2Block.FUSE_DOWN = False
    try:
        pb = PlanBuilder(n)
        blk.emit(pb, pb.new_buf(hw[0], hw[1], 32).view())
        assert L.OP_SHUFDOWN not in [op.kind for op in pb.ops]
        four = _run_yolo_block(blk, x, dev)
    finally:
        Y.ShuffleV2Block.FUSE_DOWN = True
    assert np.abs(got[:, :128] - four[:, :128]).max() <= 2e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("hw,n", [((11, 13), 3), ((32, 48), 2), ((1, 1), 2), ((80, 80), 5), ((9, 33), 4)])
def test_yolo_shufflev2_stride1_block_in_one_kernel_vs_oracle(dev, hw, n):
    """FP_OP_SHUFUNIT (csrc/shufdown.hip shufunit_x6_kernel): YOLOv5n-face's 128-channel stride-1 ShuffleV2Block (chunk, branch2,
    cat and channel_shuffle in one kernel) against the oracle block and against the two-op form (pointwise conv + FP_OP_DWPW), on
    maps whose 8 x 16 tiles overhang, divide, and at the network's own 80 x 80."""
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    from oracle import yolo_ref
    rng = np.random.default_rng(hw[0] * 5 + hw[1])
    blk = Y.ShuffleV2Block(128, 128, 1)
    blk.load_state_dict(synth_state_dict(blk.state_dict(), 670 + hw[0]))
    x = rng.normal(0, 1, (n, 128) + hw).astype(np.float32)
    pb = PlanBuilder(n)
    blk.emit(pb, pb.new_buf(hw[0], hw[1], 128).view())
    assert [op.kind for op in pb.ops] == [L.OP_SHUFUNIT] and pb.ops[0].flags == L.OPF_SPLIT3
    got = _run_yolo_block(blk.to(dev), x, dev)
    with torch.no_grad():
        want = yolo_ref._shuffle_block({k: v.cpu() for k, v in blk.state_dict().items()}, "", torch.from_numpy(x), 1).numpy()
    assert got[:, :128].shape == want.shape
    np.testing.assert_array_equal(got[:, 0:128:2], x[:, :64])                   # the x1 half passes through bit for bit
    assert rel_err(got[:, :128], want) < 1e-5
    Y.ShuffleV2Block.FUSE_UNIT = False
    try:
        pb = PlanBuilder(n)
        blk.emit(pb, pb.new_buf(hw[0], hw[1], 128).view())
        assert L.OP_SHUFUNIT not in [op.kind for op in pb.ops]
        two = _run_yolo_block(blk, x, dev)
    finally:
        Y.ShuffleV2Block.FUSE_UNIT = True
    assert np.abs(got[:, :128] - two[:, :128]).max() <= 2e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("c,hw,fuse_bn", [(32, (72, 88), False), (24, (40, 136), True), (16, (64, 64), True)])
def test_yolo_stem_fused_kernel_matches_unfused_ops(dev, c, hw, fuse_bn):
    """FP_OP_YSTEM (stem_1 -> LDS -> stem_2a + maxpool, csrc/ystem.hip) against the five separate ops on ragged maps
    (tiles of 8 x 32 stem_1 pixels that overhang the map), live and folded BatchNorm, c = 32 / 24 / 16."""
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    from oracle import yolo_ref
    rng = np.random.default_rng(c)
    blk = Y.StemBlock(3, c, 3, 2)
    blk.load_state_dict(synth_state_dict(blk.state_dict(), 700 + c))
    if fuse_bn:
        for m in (blk.stem_1, blk.stem_2a, blk.stem_2b, blk.stem_3):
            m.fuse()
    blk = blk.to(dev)
    x = rng.uniform(-1, 1, (3, 3) + hw).astype(np.float32)
    outs = {}
    for flag in (True, False):
        Y.StemBlock.FUSE = flag
        try:
            outs[flag] = _run_yolo_block(blk, x, dev)[:, :c]
        finally:
            Y.StemBlock.FUSE = True
    with torch.no_grad():
        want = yolo_ref._stem({k: v.cpu() for k, v in blk.state_dict().items()}, "", torch.from_numpy(x)).numpy()
    assert rel_err(outs[True], want) < 1e-5 and rel_err(outs[False], want) < 1e-5
    assert rel_err(outs[True], outs[False]) < 2e-6


@pytest.mark.parametrize("hw,n,fuse_bn", [((72, 88), 3, False), ((8, 8), 2, True), ((640, 640), 2, True), ((36, 132), 4, False), ((4, 4), 1, True)])
def test_yolo_stem_tail_in_one_kernel_vs_oracle(dev, hw, n, fuse_bn):
    """FP_OP_YSTEM2 (csrc/ystem2.hip): stem_2b -> cat -> stem_3 of the c = 32 StemBlock in one kernel, behind FP_OP_YSTEM: the whole
    block against the oracle (common.py:58-73 restated) and against the three-op form (FP_OP_YSTEM + two convs), live and folded
    BatchNorm, on maps whose 4 x 16 output tiles overhang (18 x 22, 2 x 2, 9 x 33, 1 x 1) and at 640 x 640."""
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    from oracle import yolo_ref
    rng = np.random.default_rng(hw[0] + 3 * hw[1])
    blk = Y.StemBlock(3, 32, 3, 2)
    blk.load_state_dict(synth_state_dict(blk.state_dict(), 720 + hw[0]))
    if fuse_bn:
        for m in (blk.stem_1, blk.stem_2a, blk.stem_2b, blk.stem_3):
            m.fuse()
    blk = blk.to(dev)
    x = rng.uniform(-1, 1, (n, 3) + hw).astype(np.float32)
    outs = {}
    for flag in (True, False):
        Y.StemBlock.FUSE_TAIL = flag
        try:
            pb = PlanBuilder(n)
            blk.emit(pb, pb.new_buf(hw[0], hw[1], 4).view())
            assert ([op.kind for op in pb.ops] == [L.OP_YSTEM, L.OP_YSTEM2]) == flag
            outs[flag] = _run_yolo_block(blk, x, dev)[:, :32]
        finally:
            Y.StemBlock.FUSE_TAIL = True
    with torch.no_grad():
        want = yolo_ref._stem({k: v.cpu() for k, v in blk.state_dict().items()}, "", torch.from_numpy(x)).numpy()
    assert outs[True].shape == want.shape
    assert rel_err(outs[True], want) < 1e-5 and rel_err(outs[False], want) < 1e-5
    assert rel_err(outs[True], outs[False]) < 2e-6


@pytest.mark.parametrize("frame_hw", [(576, 1024), (360, 640), (97, 33), (640, 640), (1275, 1650)])
def test_yolo_letterbox_fused_into_stem_is_bit_exact(dev, frame_hw):
    """FP_OP_YSTEM_U8 (the stem reads the u8 frames through fp_letterbox_tables; no fp32 canvas) against the stand-alone
    letterbox kernel + FP_OP_YSTEM: same fixed-point resize, same MFMA order -> identical decoded predictions; the
    stand-alone kernel itself is checked against the oracle's pad_resize_image in test_resize_normalize_vs_oracle."""
    from face_detection_and_recognition_amd.modules.yolov5_face import preprocess_batch
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    rng = np.random.default_rng(frame_hw[0])
    frames = torch.from_numpy(rng.integers(0, 256, (3,) + frame_hw + (3,), dtype=np.uint8)).to(dev)
    m = _yolo("yolov5n-0.5", dev, 5, True)
    zs = {}
    for flag in (True, False):
        Y.Model.FUSE_LETTERBOX = flag
        try:
            plan = preprocess_batch(m, frames, (256, 320))
            assert (plan.input is None) == flag
            zs[flag] = m.run_plan(plan).clone().cpu().numpy()
        finally:
            Y.Model.FUSE_LETTERBOX = True
    np.testing.assert_array_equal(zs[True], zs[False])


@pytest.mark.parametrize("cin,cout,hw,stride,res,n", [
    (64, 64, (40, 40), 1, True, 24),      # Bottleneck.cv2 with the shortcut (yolov5n C3)
    (92, 92, (33, 37), 1, False, 24),     # odd k-quad count in the last slab, ragged 8 x 16 tiles
    (96, 184, (40, 40), 1, False, 24),    # two 128-column chunks (the second one partial)
    (24, 24, (32, 48), 1, True, 24),      # single n tile: two alternating partial sums
    (48, 48, (40, 40), 1, True, 24),      # 16-column n tiles (16x16x4 MFMA): yolov5s C3 bottleneck, 3 tiles
    (44, 12, (35, 33), 1, False, 24),     # 16-column tiles: one partial tile, short last slab, ragged spatial tiles
    (72, 80, (32, 48), 1, False, 24),     # 16-column tiles: 5 tiles, slab of 8 channels (half a 16-channel group)
    (36, 108, (32, 40), 1, False, 24),    # 16-column tiles: 7 tiles (last one partial), slab of one k-quad
    (128, 128, (64, 72), 2, False, 24),   # stride 2: even / odd column planes
    (132, 96, (66, 70), 2, False, 24),    # stride 2, last slab of one k-quad, ragged tiles
])
def test_conv3_lds_image_kernel_vs_torch(dev, cin, cout, hw, stride, res, n):
    """csrc/conv3.hip (3x3 conv with the A operand read from an LDS image) against torch-CPU conv2d + SiLU
    (+ Bottleneck shortcut, y5/models/common.py:76-87), writing into a channel slice of a wider buffer."""
    import torch.nn.functional as F
    rng = np.random.default_rng(cin + cout)
    H, W = hw
    OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = rng.normal(0, 1, (n, cin, H, W)).astype(np.float32)
    w = rng.normal(0, np.sqrt(2.0 / (9 * cin)), (cout, cin, 3, 3)).astype(np.float32)
    b = rng.normal(0, 0.1, (cout,)).astype(np.float32)
    sc = rng.uniform(0.8, 1.2, (cout,)).astype(np.float32)
    pb = PlanBuilder(n)
    xin = pb.new_buf(H, W, cin + 8)                       # input is a channel slice too (in_ld > Cin)
    out = pb.new_buf(OH, OW, cout + 12)
    from face_detection_and_recognition_amd.plan import View
    xv = View(xin, 4, cin)
    kw = dict(stride=stride, pad=(1, 1), scale=sc, bias=b, act=L.ACT_SILU)
    if res:
        assert stride == 1 and cin == cout
        kw.update(res=xv, res_mode=L.RES_ADD_AFTER_ACT)
    PlanBuilder.X6 = False                                # the fp32-MFMA kernel is what this test is about
    try:
        pb.conv(xv, w, View(out, 8, cout), **kw)
    finally:
        PlanBuilder.X6 = True
    plan = CompiledPlan(pb, dev)
    assert plan.kernel_name(0).startswith("conv3_kernel"), plan.kernel_name(0)
    plan.arena.zero_()
    plan.buf_tensor(xin, n)[..., 4:4 + cin].copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1))
    plan.run()
    torch.cuda.synchronize()
    got = plan.buf_tensor(out, n)[..., 8:8 + cout].permute(0, 3, 1, 2).cpu().numpy()
    with torch.no_grad():
        y = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, stride=stride, padding=1)
        y = F.silu(y * torch.from_numpy(sc).view(1, -1, 1, 1) + torch.from_numpy(b).view(1, -1, 1, 1))
        if res:
            y = y + torch.from_numpy(x)
    assert rel_err(got, y.numpy()) < 2e-6
    # nothing outside the slice was written
    full = plan.buf_tensor(out, n).cpu().numpy()
    assert np.all(full[..., :8] == 0) and np.all(full[..., 8 + cout:] == 0)


def test_yolo_concat_in_place_matches_copies(dev):
    """Concat inputs produced straight into the concat buffer (Model._emit) vs explicit copies: identical outputs."""
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    g = golden("yolov5s_forward")
    zs = {}
    for flag in (True, False):
        Y.Model.CONCAT_IN_PLACE = flag
        try:
            m = _yolo("yolov5s", dev, int(g["seed"]), True)
            n_copy = sum(1 for op in m._emit(1, 128, 128)[0].ops if op.kind == L.OP_COPY)
            assert n_copy == (0 if flag else 8)
            zs[flag] = m(torch.from_numpy(g["x"]))[0].cpu().numpy()
        finally:
            Y.Model.CONCAT_IN_PLACE = True
    np.testing.assert_array_equal(zs[True], zs[False])


def test_dets_to_crops_yolo_rows_vs_reference_golden(dev, lib):
    """fp_dets_to_crops fmt = 1 against get_bboxes_confs_areas run by the reference itself
    (tests/golden/yolo_bboxes_confs_areas.npz): kept rows, rounded boxes and percent areas bit-exact; the crop
    rectangles against the oracle's crop arithmetic (extract_faces_from_dataset.py:289-303)."""
    from face_detection_and_recognition_amd import _lib as L
    from face_detection_and_recognition_amd.pipeline import FACE_OFFSETS, scale_coords_params
    g = golden("yolo_bboxes_confs_areas")
    dets = g["dets"]
    n = dets.shape[0]
    # two images: the golden rows, and the same rows reversed (ordering by (frame, detection))
    d = torch.from_numpy(np.stack([dets, dets[::-1].copy()])).to(dev)
    counts = torch.tensor([n, n], dtype=torch.int32, device=dev)
    cap = 2 * n
    items = torch.empty((cap, 9), dtype=torch.int32, device=dev)
    info = torch.empty((cap, 7), dtype=torch.float32, device=dev)
    nf = torch.empty((1,), dtype=torch.int32, device=dev)
    gain, px, py = scale_coords_params((640, 640), (1024, 576))
    tx, ty, bx, by = FACE_OFFSETS
    L.check(lib.fp_dets_to_crops(L.ptr(d), L.ptr(counts), 2, n, 16, 1, 640, 640, 1024, 576, 0.7, 0.12, float(gain),
                                 float(px), float(py), tx, ty, bx, by, 112, 112, cap, L.ptr(items), L.ptr(info),
                                 L.ptr(nf), L.current_stream(dev)), "fp_dets_to_crops")
    torch.cuda.synchronize()
    k = len(g["boxes"])
    assert int(nf) == 2 * k and k > 0
    info, items = info.cpu().numpy(), items.cpu().numpy()
    conf_pass = dets[:, 4] > 0.7
    areas_kept = g["areas"][g["areas"] > 0.12]                # the reference returns the unfiltered percent column
    assert len(areas_kept) == k and conf_pass.sum() == len(g["areas"])
    np.testing.assert_array_equal(info[:k, 0], 0)
    np.testing.assert_array_equal(info[:k, 1:5], g["boxes"])
    np.testing.assert_array_equal(info[:k, 5], g["confs"])
    np.testing.assert_array_equal(info[:k, 6], areas_kept)
    np.testing.assert_array_equal(info[k:2 * k, 1:5], g["boxes"][::-1])
    frame = np.zeros((576, 1024, 3), np.uint8)
    for i, box in enumerate(g["boxes"]):
        _, (x, y, xw, yh) = image_ref.crop_face(frame, box)
        assert tuple(items[i, :5]) == (0, x, y, xw - x, yh - y)


def test_yolo_decode_and_wnms_vs_reference_golden(dev):
    from face_detection_and_recognition_amd.modules.yolov5_face.general import (conv_strides_to_anchors,
                                                                                   w_non_max_suppression)
    g = golden("yolo_decode_wnms")
    z = conv_strides_to_anchors([g[f"head{i}"] for i in range(3)], dev)
    torch.cuda.synchronize()
    np.testing.assert_allclose(z.cpu().numpy(), g["z"], rtol=2e-6, atol=2e-6)        # expf vs torch sigmoid
    out = w_non_max_suppression(torch.from_numpy(g["pred"]).to(dev), 1, 0.4, 0.3)
    for i, o in enumerate(out):
        np.testing.assert_array_equal(o.cpu().numpy(), g[f"wnms{i}"])                # bit-exact keep set + rows


def test_yolo_nms_face_keep_indices_bit_exact(dev):
    from face_detection_and_recognition_amd.modules.yolov5_face.general import nms_face_device
    from oracle import yolo_ref
    rng = np.random.default_rng(8)
    B, n = 5, 3000
    pred = np.zeros((B, n, 16), np.float32)
    pred[..., 0:2] = np.round(rng.uniform(0, 640, (B, n, 2)) / 24) * 24 + rng.normal(0, 2, (B, n, 2))
    pred[..., 2:4] = rng.uniform(16, 120, (B, n, 2))
    pred[..., 4] = rng.uniform(0, 1, (B, n))
    pred[..., 5:15] = rng.uniform(0, 640, (B, n, 10))
    pred[..., 15] = rng.uniform(0.3, 1, (B, n))
    pred[4, :, 4] = 0.0                                         # image with no candidates
    pred[3, 10:, 4] = 0.0                                       # image with a handful
    out, cnt, keep, over = nms_face_device(torch.from_numpy(pred).to(dev), 0.4, 0.5)
    torch.cuda.synchronize()
    ref_out, ref_idx = yolo_ref.non_max_suppression_face(pred, 0.4, 0.5)
    assert over.cpu().numpy().sum() == 0
    for i in range(B):
        k = int(cnt[i])
        assert k == len(ref_idx[i])
        np.testing.assert_array_equal(keep[i, :k].cpu().numpy(), ref_idx[i].numpy())     # bit-exact kept indices
        np.testing.assert_array_equal(out[i, :k].cpu().numpy(), ref_out[i].numpy())


def test_yolo_nms_face_exact_threshold_ious_and_coincident_boxes(dev):
    """Centres and sizes on an 8-pixel grid: boxes coincide, nest and abut, and many pairs have an IoU of EXACTLY 0.5, 0.25 or
    1/3 -- the `iou > iou_thres` decision (general.py:370-453 through torchvision.ops.nms, restated in the oracle) sits on the
    comparison's edge for iou_thres = 0.5 and 0.25.  Scores distinct.  Kept indices and rows bit-exact against the oracle."""
    from face_detection_and_recognition_amd.modules.yolov5_face.general import nms_face_device
    from oracle import yolo_ref
    rng = np.random.default_rng(91)
    B, n = 6, 400
    pred = np.zeros((B, n, 16), np.float32)
    pred[..., 0:2] = rng.integers(4, 40, (B, n, 2)) * 8.0                     # centres
    pred[..., 2:4] = rng.choice([16.0, 32.0, 48.0, 64.0], (B, n, 2))          # widths / heights
    for b in range(B):
        pred[b, :, 4] = 0.45 + rng.permutation(n) / (2.0 * n)                 # distinct objectness, all candidates
    pred[..., 5:15] = rng.integers(0, 80, (B, n, 10)) * 8.0
    pred[..., 15] = 1.0
    pred[5, 7:, 4] = 0.0
    for iou_thres in (0.5, 0.25):
        out, cnt, keep, over = nms_face_device(torch.from_numpy(pred).to(dev), 0.4, iou_thres)
        torch.cuda.synchronize()
        ref_out, ref_idx = yolo_ref.non_max_suppression_face(pred, 0.4, iou_thres)
        assert over.cpu().numpy().sum() == 0
        for i in range(B):
            k = int(cnt[i])
            assert k == len(ref_idx[i]), (iou_thres, i)
            np.testing.assert_array_equal(keep[i, :k].cpu().numpy(), ref_idx[i].numpy())
            np.testing.assert_array_equal(out[i, :k].cpu().numpy(), ref_out[i].numpy())
    # the edge is really hit: pairs with IoU == 0.5 exactly exist among image 0's boxes
    b = pred[0]
    x1, y1, x2, y2 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    iw = np.clip(np.minimum(x2[:, None], x2[None]) - np.maximum(x1[:, None], x1[None]), 0, None)
    ih = np.clip(np.minimum(y2[:, None], y2[None]) - np.maximum(y1[:, None], y1[None]), 0, None)
    area = (x2 - x1) * (y2 - y1)
    iou = iw * ih / (area[:, None] + area[None] - iw * ih)
    assert (iou == 0.5).sum() > 50 and (iou == 0.25).sum() > 50


def test_yolo_nms_full_size_properties(dev):
    """BASELINE configs[2] size: 256 images x 25 200 rows with ~100 candidates each (clustered boxes so that NMS really
    suppresses).  Size-independent properties of non_max_suppression_face (general.py:370-453) on every image, and the
    exact kept-index list against the oracle on a sample of images."""
    from face_detection_and_recognition_amd.modules.yolov5_face.general import nms_face_device
    from oracle import yolo_ref
    g = torch.Generator(device=dev).manual_seed(3)
    B, n = 256, 25200
    pred = torch.zeros((B, n, 16), device=dev)
    centres = torch.rand((B, 12, 2), device=dev, generator=g) * 560 + 40          # 12 face-like clusters per image
    which = torch.randint(0, 12, (B, n), device=dev, generator=g)
    pred[..., 0:2] = torch.gather(centres, 1, which.unsqueeze(-1).expand(B, n, 2)) + \
        torch.randn((B, n, 2), device=dev, generator=g) * 6
    pred[..., 2:4] = torch.rand((B, n, 2), device=dev, generator=g) * 30 + 40
    pred[..., 4] = torch.rand((B, n), device=dev, generator=g) * 0.4               # objectness below the threshold ...
    hot = torch.rand((B, n), device=dev, generator=g) < 100.0 / n                  # ... except ~100 rows per image
    pred[..., 4] = torch.where(hot, 0.45 + 0.5 * torch.rand((B, n), device=dev, generator=g), pred[..., 4])
    pred[..., 5:15] = torch.rand((B, n, 10), device=dev, generator=g) * 640
    pred[..., 15] = 0.9 + 0.1 * torch.rand((B, n), device=dev, generator=g)
    out, cnt, keep, over = nms_face_device(pred, 0.4, 0.5)
    torch.cuda.synchronize()
    assert int(over.sum()) == 0
    cnt_h, keep_h, out_h, pred_h = cnt.cpu().numpy(), keep.cpu().numpy(), out.cpu().numpy(), pred.cpu().numpy()
    cand = ((pred_h[..., 4] > 0.4) & (pred_h[..., 4] * pred_h[..., 15] > 0.4)).sum(1)
    assert cand.mean() > 60 and np.all(cnt_h <= cand) and np.all(cnt_h >= 1) and cnt_h.mean() < 0.6 * cand.mean()
    for i in range(B):
        k = cnt_h[i]
        rows = out_h[i, :k]
        assert np.all(np.diff(rows[:, 4]) <= 0)                                     # score-descending
        assert len(set(keep_h[i, :k].tolist())) == k                                # no row kept twice
        assert np.all(pred_h[i, keep_h[i, :k], 4] > 0.4)                            # kept rows were candidates
        x1, y1, x2, y2 = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
        iw = np.clip(np.minimum(x2[:, None], x2[None]) - np.maximum(x1[:, None], x1[None]), 0, None)
        ih = np.clip(np.minimum(y2[:, None], y2[None]) - np.maximum(y1[:, None], y1[None]), 0, None)
        inter = iw * ih
        area = (x2 - x1) * (y2 - y1)
        iou = inter / (area[:, None] + area[None] - inter)
        np.fill_diagonal(iou, 0)
        assert iou.max() <= 0.5 + 1e-6                                              # kept boxes do not suppress each other
    for i in (0, 17, 101, 255):                                                     # exact kept rows on a sample
        ref_out, ref_idx = yolo_ref.non_max_suppression_face(pred_h[i:i + 1], 0.4, 0.5)
        np.testing.assert_array_equal(keep_h[i, :cnt_h[i]], ref_idx[0].numpy())
        np.testing.assert_array_equal(out_h[i, :cnt_h[i]], ref_out[0].numpy())


def test_cosine_filter_full_size_properties(dev):
    """BASELINE configs[4] per-GPU share (125 k gallery rows x 10 k reference rows x 512) and the full 1 M x 10 k
    problem: best in [-1, 1], keep == (best >= tau), arg in range, planted matches found, spot rows against the oracle."""
    g = torch.Generator(device=dev).manual_seed(42)
    Nr, D, tau = 10000, 512, 0.3
    R = torch.randn((Nr, D), device=dev, generator=g)
    Rh = R.cpu().numpy()
    for M in (125_000, 1_000_000):
        G = torch.randn((M, D), device=dev, generator=g)
        planted = torch.arange(0, M, max(M // 64, 1), device=dev)[:64]
        tgt = (planted * 7919) % Nr
        G[planted] = 3.0 * R[tgt] + 0.05 * torch.randn((len(planted), D), device=dev, generator=g)
        best, arg, keep = S.cosine_filter(G, R, tau)
        torch.cuda.synchronize()
        assert float(best.max()) <= 1.0 + 1e-5 and float(best.min()) >= -1.0 - 1e-5
        assert torch.equal(keep, best >= tau)
        assert int(arg.min()) >= 0 and int(arg.max()) < Nr
        assert torch.equal(arg[planted].long(), tgt) and bool((best[planted] > 0.99).all())
        rows = torch.cat([planted[:8], torch.tensor([0, 1, M // 2, M - 1], device=dev)])
        rb, ra, rk, _ = similarity_ref.cosine_filter(G[rows].cpu().numpy(), Rh, tau)
        np.testing.assert_allclose(best[rows].cpu().numpy(), rb, atol=1e-5)
        assert np.array_equal(arg[rows].cpu().numpy(), ra) or \
            np.all(np.abs(best[rows].cpu().numpy() - rb) < 1e-5)
        # without planted rows random 512-d vectors never reach tau against 10 k references
        assert int(keep.sum()) == len(planted)
        del G


def test_full_size_config1_step_sampled_frames_vs_oracle(dev):
    """BASELINE configs[1] at its full size: one FacePipeline.step on a bench batch (256 synthetic 576 x 1024 frames,
    the bench's detector calibration: ~60 candidates and ~2 faces per frame; ~520 crops through the embedder on a
    768-crop plan; cosine filter against 10 k references).  Eight sampled frames are re-done by the oracle: crop
    rectangles and face counts exact, embeddings within the north_star's 1e-4; whole batch: every embedding unit-norm,
    rows ordered by frame, counts consistent, best/arg/keep consistent with the scores."""
    from face_detection_and_recognition_amd import workload as W
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    frames = W.make_frames(256, dev, seed=1234)
    det = W.build_detector(dev, W.make_frames(64, dev, seed=999))
    emb = W.build_embedder(dev)
    ref = W.make_reference(10000, dev)
    pipe = FacePipeline(det, emb, ref, tau=0.3)
    out = pipe.step(frames)
    torch.cuda.synchronize()
    n = out["n_faces"]
    assert 256 < n < 1024
    plan = pipe.emb_plan
    assert plan.N % 256 == 0 and plan.N >= n and L.OP_DWBLOCK in [plan.ops[i].kind for i in range(plan.n_ops)]
    names = [det.net.last_plan.kernel_name(i) for i in range(det.net.last_plan.n_ops)]
    assert names[0] == "stem5_u8_x6_kernel" and names.count("blazepair_kernel<128>") == 3
    e = out["emb"].cpu().numpy()
    info = out["info"].cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)
    assert (np.diff(info[:, 0]) >= 0).all() and info[:, 0].max() < 256
    best, arg, keep = (out[k].cpu().numpy() for k in ("best", "arg", "keep"))
    assert best.shape == (n,) and (keep == (best >= 0.3)).all() and arg.min() >= 0 and arg.max() < 10000
    refn = ref.cpu().numpy()
    refn = refn / np.linalg.norm(refn, axis=1, keepdims=True)
    rows = np.random.default_rng(0).choice(n, 16, replace=False)
    np.testing.assert_allclose(best[rows], (e[rows] @ refn.T).max(1), rtol=0, atol=1e-4)
    # oracle on 8 sampled frames
    sd_det = {k: v.detach().cpu() for k, v in det.net.state_dict().items()}
    sd_emb = {k: v.detach().cpu() for k, v in emb.state_dict().items()}
    iw, ih = det.input_size
    items = out["items"].cpu().numpy()
    worst = 0.0
    for fi in (0, 1, 37, 100, 128, 201, 254, 255):
        f = frames[fi].cpu().numpy()
        lb = image_ref.pad_resize_image(f, (iw, ih))[..., ::-1].copy()
        x = torch.from_numpy(lb).permute(2, 0, 1).unsqueeze(0)
        with torch.no_grad():
            faces, _ = blazeface_ref.predict_on_batch(sd_det, x, det.net.anchors.cpu(), True)
        d = faces[0].numpy()
        mine = np.nonzero(info[:, 0] == fi)[0]
        if len(d) == 0:
            assert len(mine) == 0
            continue
        d = d[:, [1, 0, 3, 2] + list(range(4, 17))]
        post = image_ref.dets_to_boxes(d.copy(), (f.shape[1], f.shape[0]), (iw, ih), det.det_thres, det.bbox_area_thres)
        assert len(post["boxes"]) == len(mine), (fi, len(post["boxes"]), len(mine))
        for k, box in enumerate(post["boxes"]):
            crop, rect = image_ref.crop_face(f, box)
            it = items[mine[k]]
            assert (int(it[1]), int(it[2]), int(it[1] + it[3]), int(it[2] + it[4])) == tuple(int(v) for v in rect), (fi, k)
            face = image_ref.mfn_lut()[image_ref.resize_bilinear_u8(crop, (112, 112))]
            xin = torch.from_numpy(np.ascontiguousarray(face.transpose(2, 0, 1))).unsqueeze(0)
            with torch.no_grad():
                er = mobilefacenet_ref.forward(sd_emb, xin)[0].numpy()
            worst = max(worst, float(np.abs(e[mine[k]] - er).max()))
    assert worst < 1e-4, worst


def test_yolo_pipeline_matches_oracle_end_to_end(dev):
    """detect_face_yolov5_face path on a 576x1024 frame: letterbox -> yolov5n -> decode -> NMS, vs the oracle."""
    from face_detection_and_recognition_amd.modules.yolov5_face import inference_pytorch_model_yolov5_face
    from face_detection_and_recognition_amd.modules.yolov5_face.model import YOLOV5FaceModel
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import SPECS
    from oracle import yolo_ref
    m = _yolo("yolov5n", dev, 77, True)
    rng = np.random.default_rng(2)
    frame = rng.integers(0, 256, (576, 1024, 3), dtype=np.uint8)
    lb = image_ref.pad_resize_image(frame[..., ::-1], (640, 640))
    x = torch.from_numpy(image_ref.yolo_lut()[lb]).permute(2, 0, 1).unsqueeze(0)
    # calibrate the objectness bias (random weights) so that ~80 of the 25200 anchors pass conf 0.4
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        z0, _ = yolo_ref.forward(SPECS["yolov5n"], sd, x)
        obj = z0[0, :, 4].clamp(1e-6, 1 - 1e-6)
        logit = torch.log(obj / (1 - obj))
        delta = float(np.log(0.4 / 0.6) - torch.sort(logit, descending=True)[0][80]) + 0.05
        for conv in m.model[-1].m:
            conv.bias.view(3, 16)[:, 4] += delta
            conv.bias.view(3, 16)[:, 15] += 6.0
    m._plans.clear()
    model = YOLOV5FaceModel(m, 0.4, 0.0, inference_pytorch_model_yolov5_face, (640, 640))
    dets = model(frame)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        z, _ = yolo_ref.forward(SPECS["yolov5n"], sd, x)
    ref, _ = yolo_ref.non_max_suppression_face(z, 0.4, 0.5)
    ref = ref[0].numpy()
    assert len(dets) == len(ref) and len(ref) > 0
    np.testing.assert_allclose(dets[:, :4] * 640, ref[:, :4], rtol=0, atol=2e-2)
    np.testing.assert_allclose(dets[:, 4], ref[:, 4], rtol=0, atol=1e-4)


def test_full_size_config2_yolov5n_256_frames_sampled_vs_oracle(dev):
    """BASELINE configs[2] at its full size: YOLOv5n-face on 256 synthetic 576 x 1024 frames,
    letterboxed to 640 x 640 inside the stem, Detect decode, batched NMS -- the plan, kernel selection (batch-dependent:
    pwx6 tile forms, conv3 thresholds) and arena of the measured config, not of a 1-4 frame test.  Three sampled frames
    are re-done by the oracle network (y5/models/yolo.py:177-198): decoded rows within 1e-4 of the row scale (boxes: 1e-4
    of the 640-pixel image; scores 1e-4); the kept rows of every sampled frame are EXACT against the oracle NMS
    (y5/utils/general.py:370-453) run on the device's own decoded predictions."""
    from face_detection_and_recognition_amd import workload as W
    from face_detection_and_recognition_amd.modules.yolov5_face import nms_face_device, preprocess_batch
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import SPECS
    from oracle import yolo_ref
    B = 256
    frames = W.make_frames(B, dev, seed=5)
    det = W.build_yolo_detector(dev, W.make_frames(16, dev, seed=6), "yolov5n", cand_per_frame=80)
    plan = preprocess_batch(det.net, frames, (640, 640))
    names = [plan.kernel_name(i) for i in range(plan.n_ops)]
    assert plan.N == B and names[0].startswith("ystem_kernel") and any(nm.startswith("pwx6_kernel") for nm in names)
    z = det.net.run_plan(plan)
    out, cnt, keep, over = nms_face_device(z, 0.4, 0.5)
    torch.cuda.synchronize()
    assert z.shape == (B, 25200, 16) and bool(torch.isfinite(z).all()) and int(over.sum()) == 0
    cnt_h = cnt.cpu().numpy()
    cand = ((z[..., 4] > 0.4) & (z[..., 4] * z[..., 15] > 0.4)).sum(1).cpu().numpy()
    assert 30 < cand.mean() < 300 and np.all(cnt_h <= cand) and cnt_h.mean() >= 1
    sd = {k: v.detach().cpu() for k, v in det.net.state_dict().items()}
    for fi in (0, 131, 255):
        f = frames[fi].cpu().numpy()
        lb = image_ref.pad_resize_image(f[..., ::-1], (640, 640))
        x = torch.from_numpy(image_ref.yolo_lut()[lb]).permute(2, 0, 1).unsqueeze(0)
        with torch.no_grad():
            z_ref, _ = yolo_ref.forward(SPECS["yolov5n"], sd, x)
        zi, zr = z[fi].cpu().numpy(), z_ref[0].numpy()
        assert rel_err(zi, zr) < 1e-4
        np.testing.assert_allclose(zi[:, :4], zr[:, :4], rtol=0, atol=1e-4 * 640)          # boxes: pixels of the 640 canvas
        np.testing.assert_allclose(zi[:, 5:15], zr[:, 5:15], rtol=1e-4, atol=1e-4 * 640)     # landmarks (linear heads)
        np.testing.assert_allclose(zi[:, [4, 15]], zr[:, [4, 15]], rtol=0, atol=1e-4)      # objectness, class score
        ref_out, ref_idx = yolo_ref.non_max_suppression_face(zi[None], 0.4, 0.5)
        k = cnt_h[fi]
        assert k == len(ref_idx[0]) and k >= 1
        np.testing.assert_array_equal(keep[fi, :k].cpu().numpy(), ref_idx[0].numpy())       # kept row indices: exact
        np.testing.assert_array_equal(out[fi, :k].cpu().numpy(), ref_out[0].numpy())


def test_full_size_config3_yolov5s_to_1024_crops_sampled_vs_oracle(dev):
    """BASELINE configs[3] at its full size: YOLOv5s-face on 256 frames -> NMS -> fmt = 1 crops -> Mobile-FaceNet on >= 1024
    crops in ONE embedder run (a 1280-crop arena; Depth_Wise kernel choice and tile rounds of that batch), through
    FacePipeline as bench.py's `other_configs` leg runs it.  Sampled frames: boxes / crop rectangles exact against the oracle
    post-processing on the device's decoded predictions, the frame's embeddings within 1e-4 of the oracle Mobile-FaceNet
    on the oracle's crops; whole batch: unit-norm embeddings, rows ordered by frame."""
    from face_detection_and_recognition_amd import workload as W
    from face_detection_and_recognition_amd.modules.yolov5_face import preprocess_batch
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    from oracle import yolo_ref
    B = 256
    frames = W.make_frames(B, dev, seed=5)
    emb = W.build_embedder(dev)
    n, pipe, out = 0, None, None
    for cand in (6, 8, 12):                         # the candidate count that leaves >= 1024 crops after NMS + area filter
        det = W.build_yolo_detector(dev, frames, "yolov5s", cand_per_frame=cand)
        pipe = FacePipeline(det, emb, None, max_faces_per_frame=64)
        out = pipe.step(frames)
        n = out["n_faces"]
        if n >= 1024:
            break
    torch.cuda.synchronize()
    assert 1024 <= n <= 4096, n
    eplan = pipe.emb_plan
    assert eplan.N >= n and eplan.N % 256 == 0 and L.OP_DWBLOCK in [eplan.ops[i].kind for i in range(eplan.n_ops)]
    info, items, got_emb = out["info"].cpu().numpy(), out["items"].cpu().numpy(), out["emb"].cpu().numpy()
    assert got_emb.shape == (n, 512) and np.isfinite(got_emb).all()
    np.testing.assert_allclose(np.linalg.norm(got_emb, axis=1), 1.0, atol=1e-5)
    assert (np.diff(info[:, 0]) >= 0).all() and info[:, 0].max() < B
    plan = preprocess_batch(det.net, frames, det.input_size)
    z = det.net.run_plan(plan)
    sd_emb = {k: v.detach().cpu() for k, v in emb.state_dict().items()}
    worst, checked = 0.0, 0
    for fi in (0, 77, 200, 255):
        f = frames[fi].cpu().numpy()
        kept, _ = yolo_ref.non_max_suppression_face(z[fi:fi + 1].cpu().numpy(), 0.4, 0.5)
        boxes, confs, areas = yolo_ref.get_bboxes_confs_areas(kept[0].numpy(), det.det_thres, det.bbox_area_thres,
                                                              (1024, 576), (640, 640))
        mine = np.nonzero(info[:, 0] == fi)[0]
        assert len(mine) == len(boxes), (fi, len(mine), len(boxes))
        np.testing.assert_array_equal(info[mine, 1:5], boxes)
        np.testing.assert_array_equal(info[mine, 5], confs)
        for j, box in enumerate(boxes):
            crop, (x, y, xw, yh) = image_ref.crop_face(f, box)
            assert tuple(items[mine[j], :5]) == (fi, x, y, xw - x, yh - y)
            face = image_ref.mfn_lut()[image_ref.resize_bilinear_u8(crop, (112, 112))]
            xin = torch.from_numpy(np.ascontiguousarray(face.transpose(2, 0, 1))).unsqueeze(0)
            with torch.no_grad():
                er = mobilefacenet_ref.forward(sd_emb, xin)[0].numpy()
            worst = max(worst, float(np.abs(got_emb[mine[j]] - er).max()))
            checked += 1
    assert checked >= 4 and worst < 1e-4, (checked, worst)

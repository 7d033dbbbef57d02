"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/facepath.h declares,
plans validate, state_dict keys match the reference's names, helpers behave.  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from face_detection_and_recognition_amd import _lib as L
from face_detection_and_recognition_amd.modules.blazeface.blazeface import BlazeBlock, BlazeFace, generate_anchors
from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import MobileFaceNet
from face_detection_and_recognition_amd.modules.utils import image as uimage
from face_detection_and_recognition_amd.modules.utils.inference import get_dets_bboxes_confs_lmarks_areas
from face_detection_and_recognition_amd.modules.utils.parser import get_argparse
from face_detection_and_recognition_amd.plan import PlanBuilder, pack_conv_weight, validate_on_host


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "facepath.h")).read()
    declared = set(re.findall(r"\b(fp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fp_status"}
    assert declared, "no declarations parsed"
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in facepath.h but not exported: {missing}"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    assert lib.fp_abi_version() == L.ABI_VERSION == int(re.search(r"#define FP_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.fp_strerror(-2).decode().startswith("op touches")


def test_library_selftest_of_integer_helpers(lib):
    """fp_selftest: the reciprocal-multiply division of the persistent kernels (csrc/common.h fp_make_divisor /
    fp_fastdiv) equals n // d for every divisor 2..4096 and a set of large ones, at quotient boundaries up to 2^31 - 1."""
    assert lib.fp_selftest() == 0, lib.fp_last_hip_error().decode()


def test_struct_layout_matches_header():
    # 22 int32 + 10 int64 + 2 int32 (act2, flags) = 176 bytes; fp_resize_item = 9 int32; fp_ext = pointer + size_t
    assert ctypes.sizeof(L.FpOp) == 22 * 4 + 10 * 8 + 4 * 4
    assert ctypes.sizeof(L.FpExt) == 16
    assert ctypes.sizeof(L.FpResizeItem) == 36


def test_header_is_plain_c_and_a_c_caller_agrees_with_the_python_mirror(lib, tmp_path):
    """The boundary is a C ABI: include/facepath.h compiles as C99 (-pedantic, no C++ in it), a C program links against
    libfacepath.so and drives the GPU-free entry points (fp_abi_version, the JPEG host half on a fixture), and the struct
    layouts the C compiler derives from the header -- sizes and the offset of every field -- are the ones _lib.py's ctypes
    mirrors use."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    mirrors = {"fp_op": L.FpOp, "fp_ext": L.FpExt, "fp_resize_item": L.FpResizeItem, "fp_jpeg_info": L.FpJpegInfo}
    prints = []
    for cname, cls in mirrors.items():
        prints.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            prints.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src = tmp_path / "caller.c"
    template = open(os.path.join(ROOT, "tests", "data", "c_caller.c.in")).read()
    src.write_text(template.replace("/*LAYOUT_PRINTS*/", "\n  ".join(prints)))
    libdir = os.path.dirname(L.LIB_PATH)
    exe = tmp_path / "caller"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                        "-L", libdir, "-lfacepath", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    fixture = os.path.join(ROOT, "tests", "golden", "jpeg", "ref_test2_faces_3.jpg")
    r = subprocess.run([str(exe), fixture], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    got = dict(ln.rsplit(" ", 1) for ln in r.stdout.splitlines() if ln.startswith("fp_"))
    for cname, cls in mirrors.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)
    lines = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if not ln.startswith("fp_")}
    assert int(lines["abi"][0]) == L.ABI_VERSION
    # the same file through the Python binding: same geometry, same coefficients
    data = open(fixture, "rb").read()
    info = L.FpJpegInfo()
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    assert lib.fp_jpeg_parse(buf, len(data), ctypes.byref(info)) == 0
    coefs = np.zeros(int(info.n_coefs), np.int16)
    assert lib.fp_jpeg_entropy_decode(buf, len(data), ctypes.byref(info), coefs.ctypes.data_as(ctypes.c_void_p)) == 0
    want_sum = int((coefs.astype(np.int64) * (np.arange(len(coefs)) % 7 + 1)).sum())
    assert [int(v) for v in lines["jpeg"][:5]] == [info.width, info.height, info.ncomp, int(info.n_coefs), want_sum]
    assert lines["jpeg"][5:] == ["rc", "0"] and lines["err"] == ["-1", "-1"]


def test_plans_validate_and_reject_bad_offsets(lib):
    for net in (BlazeFace(True), BlazeFace(False), MobileFaceNet(512)):
        pb = net._emit(3)[0]
        assert validate_on_host(pb) == 0
        ops, weights, arena = pb.finish()
        arr = (L.FpOp * len(ops))(*ops)
        assert lib.fp_plan_validate(arr, len(ops), weights.size, arena - 1) == -2          # arena too small
        assert lib.fp_plan_validate(arr, len(ops), weights.size - 1, arena) == -2          # weight blob too small
        bad = (L.FpOp * len(ops))(*ops)
        bad[0].N = 0
        assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -1
    assert lib.fp_plan_validate(None, 0, 0, 0) == -1


def test_stem_with_depthwise_op_validation(lib):
    """FP_OPF_OUT_DW (Mobile-FaceNet's conv1 + conv2_dw in one kernel, csrc/stemdw.hip): the first op of the embedder plan in both
    arithmetic forms; the validator checks the depthwise block behind the conv's slopes and the split weight planes, and refuses
    the flag on any other shape or op kind."""
    from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import Depth_Wise
    for x6 in (True, False):
        Depth_Wise.X6 = x6
        try:
            pb = MobileFaceNet(512)._emit(3)[0]
        finally:
            Depth_Wise.X6 = True
        ops, weights, arena = pb.finish()
        op = ops[0]
        assert op.kind == L.OP_CONV and op.flags & L.OPF_OUT_DW and op.flags & L.OPF_IN_C3 and bool(op.flags & L.OPF_SPLIT3) == x6
        assert lib.fp_op_kernel_name(ctypes.byref(op)).decode() == ("stemdw_kernel<true>" if x6 else "stemdw_kernel<false>")
        assert (op.H, op.W, op.OH, op.OW, op.Cin, op.Cout, op.stride) == (112, 112, 56, 56, 4, 64, 2)
        arr = (L.FpOp * len(ops))(*ops)
        assert lib.fp_plan_validate(arr, len(ops), weights.size, arena) == 0
        bad = (L.FpOp * len(ops))(*ops)
        bad[0].slope_off = weights.size - 13 * 64 + 4                 # the depthwise block would end behind the blob
        assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -2
        bad = (L.FpOp * len(ops))(*ops)
        bad[0].act = L.ACT_RELU                                       # the kernel exists for BN + PReLU only
        assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -3
        bad = (L.FpOp * len(ops))(*ops)
        bad[1].flags |= L.OPF_OUT_DW                                  # not a CONV (invalid) / a CONV of another shape (unsupported)
        assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == (-1 if ops[1].kind != L.OP_CONV else -3)


def test_shufflev2_block_ops_validation(lib):
    """FP_OP_SHUFDOWN / FP_OP_SHUFUNIT (ABI 10; csrc/shufdown.hip): the plans the YOLOv5n-face blocks emit validate on the host; other
    shapes, an output that aliases the input, a parameter block that runs past the weights and stray flags are refused."""
    from face_detection_and_recognition_amd.modules.yolov5_face import yolo as Y
    for blk, cin, kind in ((Y.ShuffleV2Block(32, 128, 2), 32, L.OP_SHUFDOWN), (Y.ShuffleV2Block(128, 128, 1), 128, L.OP_SHUFUNIT)):
        pb = PlanBuilder(2)
        blk.emit(pb, pb.new_buf(24, 36, cin).view())
        assert [op.kind for op in pb.ops] == [kind] and validate_on_host(pb) == 0
        ops, weights, arena = pb.finish()

        def rc_with(**kw):
            op = L.FpOp.from_buffer_copy(pb.ops[0])
            for k, v in kw.items():
                setattr(op, k, v)
            return lib.fp_plan_validate((L.FpOp * 1)(op), 1, len(weights), arena)
        assert rc_with() == 0
        assert rc_with(Cmid=32) == -3 and rc_with(Cin=64) in (-3, -1) and rc_with(flags=0) == -3 and rc_with(act2=L.ACT_NONE) == -3
        assert rc_with(w_off=len(weights) - 64) == -2                        # the parameter block would run past the weights
        assert rc_with(w_off=pb.ops[0].w_off + 2) == -3                      # 16-byte alignment of the block
        if kind == L.OP_SHUFUNIT:
            assert rc_with(out_off=pb.ops[0].in_off) == -3                   # in place: tiles read their neighbours' pixels
            assert rc_with(stride=2) in (-3, -1)
        else:
            assert rc_with(H=23) in (-3, -1) and rc_with(stride=1) in (-3, -1)
    # the stem's tail (FP_OP_YSTEM2): emitted behind FP_OP_YSTEM for c = 32, refused for other shapes / a missing pooled view
    stem = Y.StemBlock(3, 32, 3, 2)
    pb = PlanBuilder(2)
    stem.emit(pb, pb.new_buf(48, 64, 4).view())
    assert [op.kind for op in pb.ops] == [L.OP_YSTEM, L.OP_YSTEM2] and validate_on_host(pb) == 0
    ops, weights, arena = pb.finish()

    def rc2(**kw):
        op = L.FpOp.from_buffer_copy(pb.ops[1])
        for k, v in kw.items():
            setattr(op, k, v)
        return lib.fp_plan_validate((L.FpOp * 1)(op), 1, len(weights), arena)
    assert rc2() == 0 and rc2(res_C=16) == -3 and rc2(Cout=48) in (-3, -2, -1) and rc2(flags=0) == -3 and rc2(res_H=5) == -3
    assert rc2(w_off=len(weights) - 64) == -2 and rc2(res_off=arena - 64) == -2
    for c in (24, 16):                                                   # other widths keep the two convs
        pb = PlanBuilder(2)
        Y.StemBlock(3, c, 3, 2).emit(pb, pb.new_buf(48, 64, 4).view())
        assert L.OP_YSTEM2 not in [op.kind for op in pb.ops] and validate_on_host(pb) == 0
    # switches off -> the op-by-op forms
    for name in ("FUSE_DOWN", "FUSE_UNIT"):
        setattr(Y.ShuffleV2Block, name, False)
    try:
        for blk, cin in ((Y.ShuffleV2Block(32, 128, 2), 32), (Y.ShuffleV2Block(128, 128, 1), 128)):
            pb = PlanBuilder(2)
            blk.emit(pb, pb.new_buf(24, 36, cin).view())
            assert not {L.OP_SHUFDOWN, L.OP_SHUFUNIT} & {op.kind for op in pb.ops} and validate_on_host(pb) == 0
    finally:
        for name in ("FUSE_DOWN", "FUSE_UNIT"):
            setattr(Y.ShuffleV2Block, name, True)


def test_blazeface_plans_validate_at_any_batch(lib):
    """The band rules of the pair kernels depend on the batch (fp_blazepair_band_rows: fewer, longer bands when there are many
    images): every batch size must still give a plan the validator accepts (a 64-row band on the 64 x 64 map -- one band per
    image -- once made batches >= 1024 fail)."""
    for n in (1, 2, 5, 64, 256, 1000, 1024, 4096):
        for back in (True, False):
            pb = BlazeFace(back)._emit(n, frame_hw=(576, 1024) if back else None)[0]
            assert validate_on_host(pb) == 0, (n, back)


def test_pair_s2_in_the_blazeface_plan(lib):
    """FP_OP_BLAZEPAIR with stride = 2 (csrc/blazepairs2.hip): the single stride-1 block that ends each 24-channel stage and the
    stride-2 block behind it are one op; its output feeds the next stage row-padded; the validator checks both blocks'
    parameter spans and rejects shapes the kernel does not exist for."""
    pb = BlazeFace(True)._emit(4, frame_hw=(576, 1024))[0]
    ops, weights, arena = pb.finish()
    names = [lib.fp_op_kernel_name(ctypes.byref(op)).decode() for op in ops]
    both = L.OPF_IN_ROWPAD | L.OPF_OUT_ROWPAD
    assert names[1:5] == ["blazepair_kernel<128>"] * 3 + ["blazepair_s2_kernel<128, 24>"]
    assert names[5:9] == ["blazepair_kernel<64>"] * 3 + ["blazepair_s2_kernel<64, 48>"]
    assert [ops[i].flags for i in (4, 8)] == [both, both] and names[9] == "blazeblock_wps_kernel<48>"
    assert (ops[4].OH, ops[4].OW, ops[4].Cout, ops[4].stride) == (64, 64, 24, 2) and (ops[8].OH, ops[8].Cout) == (32, 48)
    arr = (L.FpOp * len(ops))(*ops)
    assert lib.fp_plan_validate(arr, len(ops), weights.size, arena) == 0
    bad = (L.FpOp * len(ops))(*ops)
    bad[8].bias_off = weights.size - 24 - 40        # the stride-2 block's 48 biases would end behind the blob
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -2
    bad = (L.FpOp * len(ops))(*ops)
    bad[4].Cout = 32                                # only 24 -> 24 and 24 -> 48
    bad[4].out_ld = 32
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -3
    bad = (L.FpOp * len(ops))(*ops)
    bad[4].flags &= ~L.OPF_IN_ROWPAD                # the ring is fed from a row-padded tensor
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -3


def test_row_padded_views_in_the_blazeface_plan(lib):
    """include/facepath.h FP_OPF_*: the back model keeps the tensors between its 24 -> 24 stride-1 blocks row-padded;
    the validator rejects the flags on ops that cannot honour them and views whose pads would leave the arena.  (With the
    stage-end fusion off, BlazeBlock.PAIR_S2 = False: every stage ends with a single block and a stride-2 block.)"""
    BlazeBlock.PAIR_S2 = False
    try:
        _row_padded_views_in_the_blazeface_plan(lib)
    finally:
        BlazeBlock.PAIR_S2 = True


def _row_padded_views_in_the_blazeface_plan(lib):
    pb = BlazeFace(True)._emit(4, frame_hw=(576, 1024))[0]
    ops, weights, arena = pb.finish()
    flags = [op.flags for op in ops]
    names = [lib.fp_op_kernel_name(ctypes.byref(op)).decode() for op in ops]
    both = L.OPF_IN_ROWPAD | L.OPF_OUT_ROWPAD
    # stem -> 3 pairs + 1 single 24 -> 24 block at 128 x 128 -> stride 2 -> 3 pairs + 1 single at 64 x 64 -> stride 2 -> ...
    assert names[0].startswith("stem_conv_kernel<5, 1, true>") and flags[0] == L.OPF_OUT_ROWPAD
    assert names[1:4] == ["blazepair_kernel<128>"] * 3 and flags[1:4] == [both] * 3
    assert names[4].startswith("blazeblock_wp_kernel") and flags[4] == L.OPF_IN_ROWPAD     # feeds the dense stride-2 block
    assert flags[5] == L.OPF_OUT_ROWPAD                                                     # 24 -> 24 stride 2
    assert names[6:9] == ["blazepair_kernel<64>"] * 3 and flags[6:9] == [both] * 3
    assert names[9].startswith("blazeblock_wp_kernel") and flags[9] == L.OPF_IN_ROWPAD
    i48 = [i for i, n in enumerate(names) if n.startswith("blazeblock_wps_kernel<48>")]
    assert len(i48) == 7 and i48 == list(range(11, 18))
    assert flags[10] == L.OPF_OUT_ROWPAD and flags[17] == L.OPF_IN_ROWPAD      # 24 -> 48 stride 2, last 48 -> 48 block
    # unfused stride-2 block (dense), then the seven 96 -> 96 blocks of the 16 x 16 map as ONE op on the dense map
    assert flags[18] == 0 and flags[19] == 0 and "copy4_kernel" not in names
    assert names[20] == "blazechain96_kernel" and flags[20] == L.OPF_SPLIT3 and ops[20].Cmid == 7
    bad = (L.FpOp * len(ops))(*ops)
    bad[20].Cmid = 8                               # an eighth block's parameters would lie behind this op's weights ...
    bad[20].w_off = weights.size - 7 * 15104       # ... and here behind the blob
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -2
    bad = (L.FpOp * len(ops))(*ops)
    bad[20].flags = 0                              # the chain exists only in the split-MFMA form
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -3
    net = BlazeFace(True)
    net.co_scheduled = True                        # plans that run beside another network's kernels: no whole-CU op
    assert "blazechain96_kernel" not in [lib.fp_op_kernel_name(ctypes.byref(op)).decode() for op in net._emit(4, frame_hw=(576, 1024))[0].finish()[0]]
    BlazeBlock.CHAIN = False
    try:
        names2 = [lib.fp_op_kernel_name(ctypes.byref(op)).decode() for op in BlazeFace(True)._emit(4, frame_hw=(576, 1024))[0].finish()[0]]
    finally:
        BlazeBlock.CHAIN = True
    i96 = [i for i, n in enumerate(names2) if n.startswith("blazeblock_wps_kernel<96>")]
    assert len(i96) == 7 and names2.index("copy4_kernel") == 20      # the per-block kernels read a row-padded copy
    BlazeBlock.PAIR = False
    try:
        names1 = [lib.fp_op_kernel_name(ctypes.byref(op)).decode() for op in BlazeFace(True)._emit(4, frame_hw=(576, 1024))[0].finish()[0]]
    finally:
        BlazeBlock.PAIR = True
    assert sum(n.startswith("blazeblock_wp_kernel") for n in names1) == 14 and not any("blazepair" in n for n in names1)
    arr = (L.FpOp * len(ops))(*ops)
    assert lib.fp_plan_validate(arr, len(ops), weights.size, arena) == 0
    # row-padded buffers sit behind the recycled arena and never share floats with a dense view
    lo = min(b.off - (b.W + 2) * b.C for b in pb.rowpad_bufs)
    for op in ops:
        if not op.flags & L.OPF_IN_ROWPAD and op.kind not in (L.OP_STEM_U8,):
            assert op.in_off + (op.N - 1) * op.in_ns + (op.H * op.W - 1) * op.in_ld + op.Cin <= lo
    bad = (L.FpOp * len(ops))(*ops)
    bad[10].flags |= L.OPF_IN_ROWPAD               # a 24 -> 48 stride-2 block cannot read the padded layout
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -3
    bad = (L.FpOp * len(ops))(*ops)
    bad[1].in_off = 8                              # the pad row above image 0 would start before the arena
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -2
    bad = (L.FpOp * len(ops))(*ops)
    bad[1].flags = 64                              # unknown flag bit
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -1
    bad = (L.FpOp * len(ops))(*ops)
    bad[1].flags |= L.OPF_IN_C3                    # "fourth channel is padding" only makes sense on a 4-float pixel
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -1
    # the fp32-canvas form of the stem is a CONV on 4-float pixels with 3-channel weights: the planner says so
    pb2 = BlazeFace(True)._emit(4)[0]
    assert pb2.ops[0].kind == L.OP_CONV and pb2.ops[0].flags & L.OPF_IN_C3 and validate_on_host(pb2) == 0


def test_null_and_size_argument_checks(lib):
    assert lib.fp_blaze_decode(None, None, None, 1, 896, 1., 1., 1., 1., 100., .5, None, None, None) == -1
    assert lib.fp_cosine_filter(None, None, 1, None, None, 1, 512, 0., None, None, None, None, None) == -1
    assert lib.fp_yolo_nms_scratch_bytes(2, 1024) == 2 * 1024 * 32
    assert lib.fp_yolo_nms_scratch_bytes(0, 1024) == 0


def test_weight_packing_layout():
    w = np.arange(2 * 3 * 1 * 1, dtype=np.float32).reshape(2, 3, 1, 1)     # Cout=2, Cin=3
    p = pack_conv_weight(w, 4, 2).reshape(2, 32, 4)                        # Kpad=8 -> 2 quads, Npad=32
    for k in range(3):
        for n in range(2):
            assert p[k // 4, n, k % 4] == w[n, k, 0, 0]
    assert p[:, 2:, :].sum() == 0 and p[0, :, 3].sum() == 0


def test_anchor_generator_and_state_dict_keys():
    a = generate_anchors(True)
    assert a.shape == (896, 4) and np.all(a[:, 2:] == 1)
    np.testing.assert_allclose(a[0, :2], [1 / 32, 1 / 32])
    np.testing.assert_allclose(a[512, :2], [1 / 16, 1 / 16])
    keys = list(BlazeFace(True).state_dict())
    assert keys[:4] == ["backbone.0.weight", "backbone.0.bias", "backbone.2.convs.0.weight", "backbone.2.convs.0.bias"]
    assert "final.convs.1.bias" in keys and "regressor_16.weight" in keys
    mk = list(MobileFaceNet(512).state_dict())
    for k in ("conv1.conv.weight", "conv1.bn.running_var", "conv1.prelu.weight", "conv_23.conv.conv.weight",
              "conv_3.model.0.conv_dw.bn.weight", "linear.weight", "bn.num_batches_tracked"):
        assert k in mk


def test_no_cpu_fallback():
    net = BlazeFace(True)
    with pytest.raises(L.FacepathError):
        net.plan_for(1)                      # parameters on CPU: refuses instead of falling back
    with pytest.raises(RuntimeError):
        net.backbone[2](torch.zeros(1, 24, 8, 8))   # containers never compute


def test_post_processing_helpers_match_reference_golden():
    g = golden("utils_postprocess")
    post = get_dets_bboxes_confs_lmarks_areas(g["blaze_dets"].copy(), (1024, 576), (256, 256), 0.7, 0.12)
    np.testing.assert_array_equal(post.boxes, g["blaze_boxes"])
    np.testing.assert_array_equal(post.bbox_lmarks, g["blaze_lmarks"])
    np.testing.assert_array_equal(post.bbox_confs, g["blaze_confs"])
    np.testing.assert_allclose(uimage.standardize_image(g["std_in"].astype(np.float64)), g["std_out"], atol=1e-6)
    assert uimage.check_img_size(630) == 640 and uimage.make_divisible(33, 32) == 64
    assert uimage.letterbox_geometry(1024, 576, 256, 256) == (256, 144, 0, 56)
    assert abs(uimage.calculate_bbox_iou((0, 0, 2, 2), (1, 1, 3, 3)) - 1 / 7) < 1e-12


def test_cli_flags():
    p = get_argparse()
    a = p.parse_args(["-i", "x.jpg", "--md", "w.pth", "--dt", "0.5", "--at", "0.2", "-d", "hip:1"])
    assert (a.input_src, a.model, a.det_thres, a.bbox_area_thres, a.device) == ("x.jpg", "w.pth", 0.5, 0.2, "hip:1")
    p.remove_argument("model")
    with pytest.raises(SystemExit):
        p.parse_args(["--md", "w"])


def test_yolo_plans_validate_and_keys():
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import Model
    for name, nkeys in (("yolov5n", 500), ("yolov5s", 374), ("yolov5n-0.5", 500)):
        m = Model(name)
        assert len(m.state_dict()) == nkeys
        pb, inp, heads, z_off, n_rows = m._emit(2, 64, 96)
        assert validate_on_host(pb) == 0
        assert n_rows == 3 * (8 * 12 + 4 * 6 + 2 * 3)
    m = Model("yolov5s")
    assert "model.0.stem_1.bn.running_var" in m.state_dict()
    m.fuse()
    assert "model.0.stem_1.conv.bias" in m.state_dict() and "model.0.stem_1.bn.weight" not in m.state_dict()
    # a fused checkpoint loads into a fresh model
    m2 = Model("yolov5s")
    m2.load_state_dict(m.state_dict())
    assert list(m2.state_dict()) == list(m.state_dict())


def test_plan_cache_is_lru_bounded_and_shares_weights():
    """the per-network plan cache must not pin one arena per batch size ever seen, and all plans of a
    network share one copy of the packed weights."""
    from face_detection_and_recognition_amd.plan import PlanCache
    cache = PlanCache(max_plans=3)
    built = []

    def make(key):
        def build(c):
            built.append(key)
            return ("plan", key)
        return build
    for k in (1, 2, 3):
        assert cache.get(k, make(k)) == ("plan", k)
    assert cache.get(1, make(1)) == ("plan", 1) and built == [1, 2, 3]          # hit: nothing rebuilt, 1 is now newest
    cache.get(4, make(4))                                                       # evicts the least recently used: 2
    assert len(cache) == 3 and 2 not in cache and 1 in cache and 3 in cache and 4 in cache
    cache.get(2, make(2))
    assert built == [1, 2, 3, 4, 2] and 3 not in cache
    w = np.arange(8, dtype=np.float32)
    d0 = cache.device_weights(w, torch.device("cpu"))
    assert cache.device_weights(w.copy(), torch.device("cpu")) is d0              # same content: one device copy
    assert cache.device_weights(w + 1, torch.device("cpu")) is not d0
    cache.clear()
    assert len(cache) == 0 and cache.device_weights(w, torch.device("cpu")) is not d0


def test_plan_cache_keys_follow_the_emit_switches(lib, monkeypatch):
    """a plan records the kernels the class-wide switches selected when it was emitted, so the cache key of all
    three networks carries every switch (plan.switch_key): flipping PlanBuilder.X6 (= Depth_Wise.X6, `bench.py --mfma fp32`)
    or Model.FOLD_UPSAMPLE after a plan exists builds ANOTHER plan -- no op of it carries FP_OPF_SPLIT3 -- and flipping
    back finds the first one again.  Mobile-FaceNet with the split kernels on: one capacity = one plan whatever n_run is
    (the op list does not depend on Depth_Wise.block_policy then)."""
    from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import Depth_Wise
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import Model
    cuda = torch.device("cuda")
    builds = []

    def stub(net, emit):
        monkeypatch.setattr(net, "_device", lambda: cuda)
        monkeypatch.setattr(net, "_build", lambda *a, **kw: builds.append(1) or emit(*a, **kw))

    net = BlazeFace(True)
    stub(net, lambda N, cache, frame_hw=None: net._emit(N, frame_hw)[0].finish()[0])
    ops1 = net.plan_for(4, (576, 1024))
    assert any(op.kind == L.OP_BLAZECHAIN and op.flags & L.OPF_SPLIT3 for op in ops1)
    Depth_Wise.X6 = False
    try:
        ops2 = net.plan_for(4, (576, 1024))
        assert ops2 is not ops1 and not any(op.flags & L.OPF_SPLIT3 for op in ops2)
    finally:
        Depth_Wise.X6 = True
    assert net.plan_for(4, (576, 1024)) is ops1
    BlazeBlock.PAIR = False
    try:
        assert not any(op.kind == L.OP_BLAZEPAIR and op.stride == 1 for op in net.plan_for(4, (576, 1024)))
    finally:
        BlazeBlock.PAIR = True

    m = Model("yolov5n")
    stub(m, lambda N, H, W, cache, frame_hw=None: m._emit(N, H, W, frame_hw)[0].finish()[0])
    y1 = m.plan_for(2, 128, 128)
    assert any(op.flags & L.OPF_IN_UP2 for op in y1) and any(op.flags & L.OPF_SPLIT3 for op in y1)
    Model.FOLD_UPSAMPLE = False
    PlanBuilder.X6 = False
    try:
        y2 = m.plan_for(2, 128, 128)
        assert not any(op.flags & (L.OPF_IN_UP2 | L.OPF_SPLIT3) for op in y2)
    finally:
        Model.FOLD_UPSAMPLE, PlanBuilder.X6 = True, True
    assert m.plan_for(2, 128, 128) is y1

    e = MobileFaceNet(512)
    stub(e, lambda N, cache, block_shapes=None: e._emit(N, block_shapes=block_shapes)[0].finish()[0])
    n0 = len(builds)
    p_small, p_large = e.plan_for(768, n_run=200), e.plan_for(768, n_run=528)    # block_policy differs: (), (7,)
    assert Depth_Wise.block_policy(200) != Depth_Wise.block_policy(528)
    assert p_small is p_large and len(builds) == n0 + 1
    Depth_Wise.X6 = False
    try:
        f_small, f_large = e.plan_for(768, n_run=200), e.plan_for(768, n_run=528)
        assert f_small is not f_large and not any(op.flags & L.OPF_SPLIT3 for op in f_small + f_large)
    finally:
        Depth_Wise.X6 = True


def test_fused_letterbox_plans_validate_on_host(lib):
    """The plans whose first op reads u8 frames (FP_OP_STEM_U8 / FP_OP_YSTEM_U8) pass the C validator; the ops that
    need external buffers are rejected by plain fp_plan_run-style validation of their shapes only when malformed."""
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import Model
    for net, kind in ((BlazeFace(True), L.OP_STEM_U8), (BlazeFace(False), L.OP_STEM_U8)):
        pb = net._emit(3, frame_hw=(576, 1024))[0]
        assert pb.ops[0].kind == kind and pb.ops[0].res_H == 576 and pb.ops[0].res_W == 1024
        assert validate_on_host(pb) == 0
    for name in ("yolov5n", "yolov5s", "yolov5n-0.5"):
        m = Model(name)
        assert m.letterbox_fusable(640, 640)
        pb = m._emit(2, 640, 640, frame_hw=(576, 1024))[0]
        assert pb.ops[0].kind == L.OP_YSTEM_U8 and validate_on_host(pb) == 0
        assert sum(1 for op in pb.ops if op.kind == L.OP_COPY) == 0                # Concat is addressing
        bad = m._emit(2, 640, 640, frame_hw=(576, 1024))[0]
        bad.ops[0].res_C = 40                                                      # stem_1 wider than the kernel handles
        assert validate_on_host(bad) != 0
    # tap-table entry point: argument validation happens before any launch
    assert lib.fp_letterbox_tables(576, 1024, 640, 640, 0, 0, 1024, 576, 0, 140, 640, 360, 125, 1, None, None) != 0
    buf = (ctypes.c_int32 * 8)()
    assert lib.fp_letterbox_tables(576, 1024, 640, 640, 0, 0, 2048, 576, 0, 140, 640, 360, 125, 1, buf, None) != 0
    assert lib.fp_letterbox_tables(576, 1024, 640, 640, 0, 0, 1024, 576, 0, 140, 640, 700, 125, 1, buf, None) != 0
    assert lib.fp_letterbox_tables(70000, 1024, 640, 640, 0, 0, 1024, 70000, 0, 0, 640, 640, 125, 1, buf, None) != 0


def test_split_stem_packing_and_validation(lib):
    """FP_OPF_SPLIT3 on FP_OP_STEM_U8 (ABI 11, stem5_u8_x6_kernel): PlanBuilder.pack_stem5_x6 lays the 24 x 3 x 5 x 5 weights out as
    [3 slabs][2 channel tiles][3 planes][16][32] bf16 with k = 16 (ky - 2 slab) + 3 kx + c -- the three planes add up to the fp32
    weight exactly, every other position (k = 15 and 31 of a slab, the sixth ky, channels 24 .. 31) is zero; the back model's plan
    carries the flag from batch 16 on (with PlanBuilder.X6 and BlazeFace.STEM_X6), validates, and the flag is refused on the
    stem of any other shape."""
    from face_detection_and_recognition_amd.plan import PlanBuilder
    w = np.random.default_rng(5).normal(0, 0.3, (24, 3, 5, 5)).astype(np.float32)
    blob = PlanBuilder.pack_stem5_x6(w)
    assert blob.dtype == np.float32 and blob.size == 3 * 2 * 3 * 16 * 32 // 2
    planes = blob.view(np.uint16).reshape(3, 2, 3, 16, 32).astype(np.uint32) << 16       # [slab][nt][plane][co % 16][k]
    rebuilt = planes.view(np.float32).sum(axis=2)                                       # h + m + l (exact in fp32: split3_bf16)
    seen = np.zeros_like(rebuilt, dtype=bool)
    for ky in range(5):
        for kx in range(5):
            for c in range(3):
                k = 16 * (ky % 2) + 3 * kx + c
                np.testing.assert_array_equal(rebuilt[ky // 2, :, :, k].reshape(32)[:24], w[:, c, ky, kx])
                seen[ky // 2, :, :, k].reshape(32)[:24] = True
    assert (rebuilt[~seen] == 0).all()
    for n, flagged in ((3, False), (16, True), (256, True)):
        pb = BlazeFace(True)._emit(n, frame_hw=(576, 1024))[0]
        assert bool(pb.ops[0].flags & L.OPF_SPLIT3) == flagged and validate_on_host(pb) == 0
    pb = BlazeFace(False)._emit(16, frame_hw=(576, 1024))[0]                                 # front model: 128 x 128 canvas
    assert not (pb.ops[0].flags & L.OPF_SPLIT3) and validate_on_host(pb) == 0
    pb.ops[0].flags |= L.OPF_SPLIT3
    assert validate_on_host(pb) == -3                                                        # FP_ERR_UNSUPPORTED
    old = BlazeFace.STEM_X6
    try:
        BlazeFace.STEM_X6 = False
        pb = BlazeFace(True)._emit(16, frame_hw=(576, 1024))[0]
        assert not (pb.ops[0].flags & L.OPF_SPLIT3) and validate_on_host(pb) == 0
    finally:
        BlazeFace.STEM_X6 = old


def test_split3_plan_layouts_and_validation():
    """FP_OPF_SPLIT3 on the host: the weight planes PlanBuilder packs for the split-MFMA ops rebuild the fp32 weights
    exactly in the layouts include/facepath.h documents, every network plan validates with the split kernels on and off,
    and the switch changes which ops carry the flag."""
    import numpy as np
    from face_detection_and_recognition_amd import _lib as L
    from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import Depth_Wise, MobileFaceNet
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import Model
    from face_detection_and_recognition_amd.plan import PlanBuilder, validate_on_host

    def planes_to_f32(u16):
        return (u16.astype(np.uint32) << 16).view(np.float32)

    # pointwise conv: [K / 32][3][Npad][32]
    rng = np.random.default_rng(0)
    w = rng.normal(0, 1, (92, 184, 1, 1)).astype(np.float32)
    pb = PlanBuilder(2)
    x, o = pb.new_buf(24, 24, 184), pb.new_buf(24, 24, 92)
    pb.conv(x.view(), w, o.view())
    op = pb.ops[0]
    assert op.flags & L.OPF_SPLIT3
    per, npad = PlanBuilder.x6_tiles(92)
    assert (per, npad) == (6, 96)
    cs = (184 + 31) // 32
    blob = np.concatenate(pb.wchunks)[op.w_off: op.w_off + cs * 3 * npad * 32 // 2].view(np.uint16).reshape(cs, 3, npad, 32)
    full = planes_to_f32(blob).sum(axis=1, dtype=np.float64).astype(np.float32)      # h + m + l, exact in fp64 -> fp32
    full = full.transpose(1, 0, 2).reshape(npad, cs * 32)
    np.testing.assert_array_equal(full[:92, :184], w.reshape(92, 184))
    assert not full[92:].any() and not full[:, 184:].any()                           # zero padding rows / columns
    assert validate_on_host(pb) == 0

    # every network, split kernels on / off
    nets = [MobileFaceNet(512), Model("yolov5n").fuse(), Model("yolov5s").fuse()]
    counts = {}
    for flag in (True, False):
        Depth_Wise.X6 = flag
        try:
            for net in nets:
                pbn = net._emit(4)[0]
                assert validate_on_host(pbn) == 0
                counts[(type(net).__name__, getattr(net, "name", ""), flag)] = sum(1 for q in pbn.ops if q.flags & L.OPF_SPLIT3)
        finally:
            Depth_Wise.X6 = True
    assert all(v == 0 for k, v in counts.items() if not k[2])
    assert counts[("MobileFaceNet", "", True)] == 18          # conv1 (stemdw_kernel<true>) + 15 Depth_Wise blocks + conv_6_sep + the Linear
    assert all(v > 0 for k, v in counts.items() if k[2])


def test_upsample_fold_plan_and_validation(lib):
    """FP_OPF_IN_UP2 (include/facepath.h): YOLOv5n-face's two Upsample + Concat pairs become operand addressing of the C3's
    merged cv1 | cv2 conv (no upsample2x op; the half-size map outlives the Concat), yolov5s' on the general split kernel;
    the validator rejects the flag where nothing can honour it and views that leave the arena."""
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import Model
    pb = Model("yolov5n")._emit(2, 640, 640, None)[0]
    ops, weights, arena = pb.finish()
    names = [lib.fp_op_kernel_name(ctypes.byref(op)).decode() for op in ops]
    up = [i for i, op in enumerate(ops) if op.flags & L.OPF_IN_UP2]
    assert len(up) == 2 and "upsample2x_kernel" not in names and all(names[i].endswith("true>") for i in up)
    for i in up:
        op = ops[i]
        assert (op.res_H, op.res_W) == (op.H // 2, op.W // 2) and op.res_C == 128 and op.res_mode == L.RES_NONE
        # nobody overwrites the half-size map between its producer and this reader
        lo, hi = op.res_off, op.res_off + (op.N - 1) * op.res_ns + (op.res_H * op.res_W - 1) * op.res_ld + op.res_C
        prod = max(j for j in range(i) if ops[j].out_off == op.res_off)
        for j in range(prod + 1, i):
            o = ops[j]
            oh, ow = (o.H, o.W) if o.kind in (L.OP_COPY, L.OP_L2NORM) else (o.OH, o.OW)
            assert o.out_off >= hi or o.out_off + (o.N - 1) * o.out_ns + (oh * ow - 1) * o.out_ld + o.out_ld <= lo, (i, j)
    arr = (L.FpOp * len(ops))(*ops)
    assert lib.fp_plan_validate(arr, len(ops), weights.size, arena) == 0
    bad = (L.FpOp * len(ops))(*ops)
    bad[up[0]].res_off = arena - 8                      # the half-size view would leave the arena
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -2
    bad = (L.FpOp * len(ops))(*ops)
    bad[up[0]].res_H += 1                               # not the half-size map of this op's input
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -3
    bad = (L.FpOp * len(ops))(*ops)
    k = next(i for i, op in enumerate(ops) if op.kind == L.OP_DWCONV)
    bad[k].flags |= L.OPF_IN_UP2                        # only the split pointwise conv reads a folded upsample
    assert lib.fp_plan_validate(bad, len(ops), weights.size, arena) == -1
    ops_s = Model("yolov5s")._emit(2, 640, 640, None)[0].finish()[0]      # widths 184 / 96: the general split kernel reads the fold
    names_s = [lib.fp_op_kernel_name(ctypes.byref(op)).decode() for op in ops_s]
    ups = [i for i, op in enumerate(ops_s) if op.flags & L.OPF_IN_UP2]
    assert "upsample2x_kernel" not in names_s and sorted(ops_s[i].res_C for i in ups) == [96, 184]
    assert all(names_s[i].startswith("convx6_kernel") for i in ups)
    Model.FOLD_UPSAMPLE = False
    try:
        names_u = [lib.fp_op_kernel_name(ctypes.byref(op)).decode() for op in Model("yolov5n")._emit(2, 640, 640, None)[0].finish()[0]]
    finally:
        Model.FOLD_UPSAMPLE = True
    assert names_u.count("upsample2x_kernel") == 2 and not any(n.endswith("true>") and n.startswith("pwx6") for n in names_u)

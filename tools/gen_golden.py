"""Generates tests/golden/*.npz from the REFERENCE ITSELF (this container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
The reference's model classes / post-processing functions are imported from /root/reference through
tools/ref_harness.py (stubs for the absent cv2 / torchvision / thop / seaborn), fed seeded synthetic
weights (face_detection_and_recognition_amd/synth.py regenerates the same weights from the seed on the
GPU box, so only inputs' seeds and the reference's OUTPUTS are stored) and seeded inputs.
The fixtures are data: seeds, small inputs and the reference's outputs.  No reference source is stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402
from face_detection_and_recognition_amd.synth import synth_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FDE = "/root/reference/face_detection_and_extraction"
torch.set_num_threads(4)
torch.manual_seed(0)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def crafted_detections(rng, n_faces, per_face, jitter, base_size=0.18):
    """(n,17) detections with w,h > 0 clustered around n_faces centres."""
    rows = []
    for f in range(n_faces):
        cx, cy = rng.uniform(0.2, 0.8, 2)
        for _ in range(per_face):
            dx, dy = rng.normal(0, jitter, 2)
            w, h = base_size * rng.uniform(0.8, 1.25, 2)
            x0, y0 = cx + dx - w / 2, cy + dy - h / 2
            kp = rng.uniform(0, 1, 12)
            rows.append([y0, x0, y0 + h, x0 + w, *kp, rng.uniform(0.65, 0.99)])
    a = np.asarray(rows, dtype=np.float32)
    return a[rng.permutation(len(a))]


def gen_blazeface(ns):
    bf = ns.blazeface
    for back in (True, False):
        tag = "back" if back else "front"
        net = bf.BlazeFace(back_model=back).eval()
        sd = synth_state_dict(net.state_dict(), seed=100 + int(back), residual_gain=0.5)
        net.load_state_dict(sd)
        S = 256 if back else 128
        rng = np.random.default_rng(7 + int(back))
        x_u8 = rng.integers(0, 256, size=(2, S, S, 3), dtype=np.uint8)          # RGB HWC
        x = torch.from_numpy(x_u8).permute(0, 3, 1, 2)
        with torch.no_grad():
            r, c = net(net._preprocess(x))
        save(f"blazeface_{tag}_forward", seed=100 + int(back), x_u8=x_u8, r=r.numpy(), c=c.numpy())

    # blocks (G1): BlazeBlock s1 24->24 @16x16, s2 24->48, FinalBlazeBlock 96 @8x8
    rng = np.random.default_rng(11)
    for name, blk, cin, hw in (("s1", bf.BlazeBlock(24, 24), 24, 16), ("s2", bf.BlazeBlock(24, 48, stride=2), 24, 16),
                               ("final", bf.FinalBlazeBlock(96), 96, 8)):
        blk.eval()
        sd = synth_state_dict(blk.state_dict(), seed=200 + len(name))
        blk.load_state_dict(sd)
        x = torch.from_numpy(rng.normal(0, 1, (2, cin, hw, hw)).astype(np.float32))
        with torch.no_grad():
            y = blk(x)
        save(f"blazeblock_{name}", seed=200 + len(name), x=x.numpy(), y=y.numpy())

    # decode + threshold (G3) and weighted NMS (G4) through the reference's own methods
    net = bf.BlazeFace(back_model=True).eval()
    rng = np.random.default_rng(21)
    anchors = rng.uniform(0.05, 0.95, (896, 4)).astype(np.float32)
    anchors[:, 2:] = 1.0
    raw_box = rng.normal(0, 20, (3, 896, 16)).astype(np.float32)
    raw_box[..., 2:4] = np.abs(raw_box[..., 2:4]) + 20.0                            # w,h > 0
    raw_score = rng.normal(-3.0, 2.5, (3, 896, 1)).astype(np.float32)
    raw_score[0, :5, 0] = [150.0, -150.0, 0.61, 0.62, 0.63]                         # clip + around-threshold cases
    net.anchors = torch.from_numpy(anchors)
    dets = net._tensors_to_detections(torch.from_numpy(raw_box), torch.from_numpy(raw_score), net.anchors)
    boxes = net._decode_boxes(torch.from_numpy(raw_box), net.anchors)
    save("blazeface_decode", anchors=anchors, raw_box=raw_box, raw_score=raw_score, boxes=boxes.numpy(),
         counts=np.array([len(d) for d in dets]), **{f"dets{i}": d.numpy() for i, d in enumerate(dets)})

    scenes = {
        "no_overlap": np.stack([[0.1 * i, 0.1 * i, 0.1 * i + 0.05, 0.1 * i + 0.05] + [0.5] * 12 + [0.7 + 0.02 * i]
                                for i in range(8)]).astype(np.float32),
        "clusters": crafted_detections(np.random.default_rng(31), 3, 12, 0.02),
        "chains": crafted_detections(np.random.default_rng(32), 2, 40, 0.06),
        "all_overlap": crafted_detections(np.random.default_rng(33), 1, 64, 0.005),
        "many": crafted_detections(np.random.default_rng(34), 12, 40, 0.03),
        "single": crafted_detections(np.random.default_rng(35), 1, 1, 0.0),
    }
    out = {}
    for k, d in scenes.items():
        faces = net._weighted_non_max_suppression(torch.from_numpy(d))
        out[k + "_in"] = d
        out[k + "_out"] = torch.stack(faces).numpy() if faces else np.zeros((0, 17), np.float32)
    save("blazeface_wnms", **out)


def gen_mobilefacenet(ns):
    m = ns.mobile_facenet.MobileFaceNet(512).eval()
    sd = synth_state_dict(m.state_dict(), seed=300)
    m.load_state_dict(sd)
    rng = np.random.default_rng(41)
    x = torch.from_numpy(rng.uniform(-1, 1, (4, 3, 112, 112)).astype(np.float32))
    with torch.no_grad():
        e = m(x)
        # intermediate taps for layer-level debugging
        t1 = m.conv2_dw(m.conv1(x))
        t2 = m.conv_3(m.conv_23(t1))
    save("mobilefacenet_forward", seed=300, x=x.numpy(), emb=e.numpy(), tap_conv2_dw=t1.numpy()[:1],
         tap_conv_3=t2.numpy()[:1])

    dw = ns.mobile_facenet.Depth_Wise(64, 64, residual=True, kernel=(3, 3), stride=(1, 1), padding=(1, 1), groups=128).eval()
    dw.load_state_dict(synth_state_dict(dw.state_dict(), seed=301))
    dn = ns.mobile_facenet.Depth_Wise(64, 128, kernel=(3, 3), stride=(2, 2), padding=(1, 1), groups=256).eval()
    dn.load_state_dict(synth_state_dict(dn.state_dict(), seed=302))
    x = torch.from_numpy(rng.normal(0, 1, (2, 64, 14, 14)).astype(np.float32))
    with torch.no_grad():
        save("mobilefacenet_depthwise", x=x.numpy(), y_res=dw(x).numpy(), y_down=dn(x).numpy())


def gen_utils(ns):
    rng = np.random.default_rng(51)
    # get_dets_bboxes_confs_lmarks_areas + scale_coords for 576x1024 -> 256^2 and 640^2 (G10)
    out = {}
    for tag, (iw, ih), K in (("blaze", (256, 256), 17), ("yolo", (640, 640), 5)):
        n = 12
        xy0 = rng.uniform(0.05, 0.6, (n, 2))
        wh = rng.uniform(0.02, 0.35, (n, 2))
        dets = np.concatenate([xy0, xy0 + wh, rng.uniform(0.1, 0.9, (n, K - 5)), rng.uniform(0.3, 1.0, (n, 1))], axis=1)
        post = ns.inference.get_dets_bboxes_confs_lmarks_areas(dets.copy(), (1024, 576), (iw, ih), 0.7, 0.12)
        out.update({f"{tag}_dets": dets, f"{tag}_boxes": post.boxes, f"{tag}_confs": post.bbox_confs,
                    f"{tag}_areas": post.bbox_areas, f"{tag}_lmarks": post.bbox_lmarks})
    img = rng.integers(0, 256, (2, 20, 24, 3)).astype(np.uint8)
    out["std_in"] = img
    out["std_out"] = ns.image.standardize_image(img.astype(np.float64))
    save("utils_postprocess", **out)


def gen_similarity():
    """S1/S2 from the reference's OWN code: similar_face_filtering/filter_faces_using_reference.py is imported on a
    tensorflow stub (tools/ref_harness.install_tensorflow_stub: images are paths, the "model" returns seeded feature
    rows keyed by file name) and both get_ref_mean_vec_and_thres_from_imgs (:71-100) and main() (:127-199) run on a
    temporary tree of three classes: a full one (40 reference files, 32 used), a short one (5) and one whose unfiltered
    images include exact copies of reference features (distance == thres is kept: `<=`).  Stored per class: the
    reference features in the order the function consumed them (glob order), its mean and threshold, the unfiltered
    features and main()'s clean / unclean decision for each.  The S3 cosine rows stay numpy (the tracker golden pins
    the reference's own cosine through Net.check_if_face_exists)."""
    import shutil
    import tempfile
    D = 512
    rng = np.random.default_rng(61)
    feats = {}
    root = tempfile.mkdtemp(prefix="sff_golden_")
    classes = [("AA-FULL", 40, 40), ("BB-SHORT", 5, 23), ("CC-EDGE", 12, 17)]
    try:
        for ci, (cls, n_ref, n_unf) in enumerate(classes):
            centre = rng.normal(0, 1, D).astype(np.float32)
            for kind, n in (("ref", n_ref), ("unf", n_unf)):
                d = os.path.join(root, kind, cls)
                os.makedirs(d)
                for k in range(n):
                    name = f"{cls}_{kind}_{k:03d}.jpg"
                    with open(os.path.join(d, name), "wb") as f:
                        f.write(b"not a jpeg: the tensorflow stub never decodes")
                    spread = 0.35 if kind == "ref" else rng.uniform(0.15, 0.75)
                    feats[name] = (centre + rng.normal(0, spread, D)).astype(np.float32)
            if cls == "CC-EDGE":                     # unfiltered images that ARE reference images (dist == thres for one)
                for k in range(6):
                    feats[f"{cls}_unf_{k:03d}.jpg"] = feats[f"{cls}_ref_{k:03d}.jpg"].copy()
        log = []
        ffr = rh.import_reference_filter(lambda p: feats[os.path.basename(p)], log)
        out = {"classes": np.array([c[0] for c in classes])}
        model = ffr.tf.keras.models.load_model("stub")
        for ci, (cls, n_ref, n_unf) in enumerate(classes):
            del log[:]
            mean, thres = ffr.get_ref_mean_vec_and_thres_from_imgs(model, os.path.join(root, "ref", cls), 32)
            used = [os.path.basename(p) for p in log]
            assert len(used) == min(32, n_ref) and mean.shape == (1, D)
            out[f"c{ci}_ref"] = np.stack([feats[u] for u in used])
            out[f"c{ci}_mean"] = np.asarray(mean[0], np.float32)
            out[f"c{ci}_thres"] = np.float32(thres)
        # main(): the reference pairs ref / unfiltered classes by glob order of the two roots; give it sorted-stable names
        tgt = os.path.join(root, "out")
        argv = sys.argv
        sys.argv = ["filter_faces_using_reference.py", "--ud", os.path.join(root, "unf"), "--rd", os.path.join(root, "ref"),
                    "--td", tgt, "-b", "7", "-r", "32"]
        try:
            import glob as _glob
            order_ref = [os.path.basename(p) for p in _glob.glob(os.path.join(root, "ref", "*"))]
            order_unf = [os.path.basename(p) for p in _glob.glob(os.path.join(root, "unf", "*"))]
            assert order_ref == order_unf, "glob order of the two roots differs on this filesystem: main() would raise"
            ffr.main()
        finally:
            sys.argv = argv
        for ci, (cls, n_ref, n_unf) in enumerate(classes):
            names = sorted(f"{cls}_unf_{k:03d}.jpg" for k in range(n_unf))
            clean = set(os.listdir(os.path.join(tgt, "clean", cls)))
            unclean = set(os.listdir(os.path.join(tgt, "unclean", cls)))
            assert clean | unclean == set(names) and not (clean & unclean)
            out[f"c{ci}_E"] = np.stack([feats[n] for n in names])
            out[f"c{ci}_keep"] = np.array([n in clean for n in names])
    finally:
        shutil.rmtree(root, ignore_errors=True)
    # legacy keys = class 0 (what the HIP tests read); dist has no reference output (main() only acts on the comparison)
    mean0, thres0 = out["c0_mean"], out["c0_thres"]
    dist = np.array([np.linalg.norm(o - mean0) for o in out["c0_E"]])
    a = rng.normal(0, 1, (64, 512)).astype(np.float32)
    b = rng.normal(0, 1, (32, 512)).astype(np.float32)
    cos = np.array([[1 - np.dot(x, y) / (np.linalg.norm(x) * np.linalg.norm(y)) for y in b] for x in a])
    save("similarity", ref=out["c0_ref"], mean=mean0, thres=thres0, E=out["c0_E"], dist=dist, keep=out["c0_keep"],
         cos_a=a, cos_b=b, cos_dist=cos, **out)


def gen_yolo():
    ys = rh.import_reference_yolo()
    rng = np.random.default_rng(71)
    for name in ("yolov5n", "yolov5s"):
        m = ys.build_model(name + ".yaml")
        sd0 = m.state_dict()
        sd = synth_state_dict(sd0, seed=400 + len(name))
        m.load_state_dict(sd)
        m = m.fuse().eval()
        x = torch.from_numpy(rng.uniform(0, 1, (1, 3, 128, 128)).astype(np.float32))
        with torch.no_grad():
            z, heads = m(x)
        save(f"{name}_forward", seed=400 + len(name), x=x.numpy(), z=z.numpy(),
             **{f"head{i}": h.numpy() for i, h in enumerate(heads)})

    # Detect decode on crafted heads at 64x64 input (G5) through the ONNX-variant decode (pure reference)
    heads = [rng.normal(0, 1.5, (2, 3, 64 // s, 64 // s, 16)).astype(np.float32) for s in (8, 16, 32)]
    z = ys.onnx_utils.conv_strides_to_anchors([h.copy() for h in heads], "cpu")
    # NMS: w_non_max_suppression is pure reference; non_max_suppression_face needs torchvision.ops.nms (absent)
    pred = z.numpy().copy()
    pred[..., 4] = rng.uniform(0, 1, pred.shape[:2]).astype(np.float32)
    pred[..., 15] = rng.uniform(0.5, 1, pred.shape[:2]).astype(np.float32)
    # make boxes cluster: snap centres to a coarse grid so IoUs are substantial
    pred[..., 0:2] = np.round(pred[..., 0:2] / 16) * 16 + rng.normal(0, 1.5, pred[..., 0:2].shape)
    pred[..., 2:4] = np.abs(pred[..., 2:4]) % 40 + 12
    pred = pred.astype(np.float32)
    wout = ys.onnx_utils.w_non_max_suppression(torch.from_numpy(pred.copy()), num_classes=1, conf_thres=0.4, nms_thres=0.3)
    save("yolo_decode_wnms", z=z.numpy(), pred=pred, **{f"head{i}": h for i, h in enumerate(heads)},
         **{f"wnms{i}": (o.numpy() if o is not None else np.zeros((0, 7), np.float32)) for i, o in enumerate(wout)})
    # box_iou (pure torch in the reference) pins the IoU formula torchvision.ops.nms uses
    b1 = rng.uniform(0, 100, (16, 2)); b1 = np.concatenate([b1, b1 + rng.uniform(5, 60, (16, 2))], 1).astype(np.float32)
    iou = ys.general.box_iou(torch.from_numpy(b1), torch.from_numpy(b1))
    save("yolo_box_iou", boxes=b1, iou=iou.numpy())


def gen_yolo_blocks():
    """Per-block fixtures of the YOLOv5-face path: Conv (+ fuse_conv_and_bn), StemBlock, ShuffleV2Block
    s1 / s2, C3 (with and without shortcut), SPP at small maps -- the reference's own module classes on seeded weights --
    plus the yolov5n-0.5 whole net (y5/models/yolov5n-0.5.yaml) and get_bboxes_confs_areas (onnx_utils.py:313-340)."""
    ys = rh.import_reference_yolo()
    C = ys.common
    rng = np.random.default_rng(81)
    out = {}

    def run(tag, mod, shape, seed):
        ys.torch_utils.initialize_weights(mod)    # as Model.__init__ does (yolo.py:152): BatchNorm eps = 1e-3
        mod = mod.eval()
        mod.load_state_dict(synth_state_dict(mod.state_dict(), seed=seed))
        x = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32))
        with torch.no_grad():
            y = mod(x)
        out[f"{tag}_seed"], out[f"{tag}_x"], out[f"{tag}_y"] = seed, x.numpy(), y.numpy()
        return mod, x

    conv, x = run("conv", C.Conv(16, 32, 3, 2), (2, 16, 32, 32), 601)
    conv.conv = ys.torch_utils.fuse_conv_and_bn(conv.conv, conv.bn)          # what Model.fuse does (yolo.py:225-233)
    with torch.no_grad():
        out["conv_y_fused"] = conv.fuseforward(x).numpy()
    run("conv1x1", C.Conv(24, 40, 1, 1), (2, 24, 20, 20), 602)
    run("stem", C.StemBlock(3, 32, 3, 2), (2, 3, 64, 64), 603)
    run("shuffle_s2", C.ShuffleV2Block(32, 128, 2), (2, 32, 16, 16), 604)
    run("shuffle_s1", C.ShuffleV2Block(128, 128, 1), (2, 128, 8, 8), 605)
    run("c3", C.C3(64, 64, 2), (2, 64, 16, 16), 606)
    run("c3_noshortcut", C.C3(96, 64, 1, False), (2, 96, 8, 8), 607)
    run("spp", C.SPP(128, 128, (3, 5, 7)), (2, 128, 8, 8), 608)
    save("yolo_blocks", **out)

    m = ys.build_model("yolov5n-0.5.yaml")
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=411))
    m = m.fuse().eval()
    x = torch.from_numpy(rng.uniform(0, 1, (1, 3, 128, 128)).astype(np.float32))
    with torch.no_grad():
        z, heads = m(x)
    save("yolov5n-0.5_forward", seed=411, x=x.numpy(), z=z.numpy(), **{f"head{i}": h.numpy() for i, h in enumerate(heads)})

    # get_bboxes_confs_areas on fp32 YOLO rows in 640^2 input pixels, frame 1024x576; rows straddle both thresholds
    n = 48
    xy0 = rng.uniform(0, 560, (n, 2))
    wh = rng.uniform(2, 260, (n, 2))
    wh[:8] = rng.uniform(18, 26, (8, 2))          # areas around 0.12 % of 640*640 (= 491.5 px^2)
    dets = np.concatenate([xy0, xy0 + wh, rng.uniform(0.3, 1.0, (n, 1)), rng.uniform(0, 640, (n, 10)),
                           np.ones((n, 1))], axis=1).astype(np.float32)
    dets[8:12, 4] = np.float32(0.7) + np.array([-1e-6, 0, 1e-6, 2e-6], np.float32)
    boxes, confs, areas = ys.onnx_utils.get_bboxes_confs_areas(dets.copy(), 0.7, 0.12, (1024, 576), (640, 640))
    save("yolo_bboxes_confs_areas", dets=dets, boxes=boxes, confs=confs, areas=areas)


def gen_jpeg():
    """tests/golden/jpeg: the reference's own test / example JPEGs (data files: fde/data/TEST/test1_faces_0.jpg,
    test2_faces_3.jpg -- the images its tests run on -- and two of fde/data/EXAMPLE *, one of them progressive) copied as INPUT
    fixtures, and the sha256 of what libjpeg-turbo (through Pillow, in this container) decodes them to: the pin of
    oracle/jpeg_ref.py and of the product's decoder (csrc/jpeg.hip)."""
    import hashlib
    import json
    import shutil
    from oracle import jpeg_ref
    src = {"ref_test1_faces_0.jpg": "data/TEST/test1_faces_0.jpg", "ref_test2_faces_3.jpg": "data/TEST/test2_faces_3.jpg",
           "ref_selfie3.jpeg": "data/EXAMPLE 2/selfie3.jpeg", "ref_selfie1_progressive.jpeg": "data/EXAMPLE 1/selfie1.jpeg"}
    out = os.path.join(OUT, "jpeg")
    os.makedirs(out, exist_ok=True)
    exp = {}
    for name, rel in src.items():
        dst = os.path.join(out, name)
        shutil.copyfile(os.path.join(FDE, rel), dst)
        a = jpeg_ref.decode_pil(open(dst, "rb").read())
        exp[name] = {"shape": list(a.shape), "sha256_rgb": hashlib.sha256(a.tobytes()).hexdigest(), "mean": round(float(a.mean()), 4)}
    json.dump(exp, open(os.path.join(out, "expected.json"), "w"), indent=1, sort_keys=True)
    print("jpeg:", {k: v["shape"] for k, v in exp.items()})


def gen_tracker():
    """Face-tracker matching: the reference's own Net.check_if_face_exists / Net.add_face
    (fde/face_extraction/extract_and_label_faces_from_dataset.py:101-121) on seeded feature / box sequences.  The module
    creates ./logs at import time, so it is imported with a scratch directory as cwd; the Net instance is made without
    __init__ (which only loads onnx / openvino model files)."""
    import contextlib
    import importlib
    import io
    import tempfile
    for name in ("openvino", "openvino.runtime", "openvino.inference_engine"):
        if name not in sys.modules:
            rh._stub(name, Core=object, IECore=object)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            mod = importlib.import_module("face_extraction.extract_and_label_faces_from_dataset")
        finally:
            os.chdir(cwd)
    out = {}
    for case, (kind, D, n_frames) in enumerate((("MOBILE_FACENET", 512, 24), ("FACE_REID_MNV3", 256, 24))):
        rng = np.random.default_rng(500 + case)
        net = object.__new__(mod.Net)
        net.feat_net_type, net.normal_thres, net.harsh_thres, net.use_bbox_iou = kind, 1., 0.72, True
        net.face_feat_bbox_age_gender_list, net.max_faceid = [], 0
        # 5 identities wandering through the frames: unit features + noise (same identity: dist ~0.3-0.9, different: ~1.4),
        # boxes drifting a few pixels per frame; distances within 1e-3 of a threshold are re-drawn
        protos = rng.normal(0, 1, (5, D)).astype(np.float32)
        protos /= np.linalg.norm(protos, axis=1, keepdims=True)
        centres = rng.uniform(100, 900, (5, 2))
        feats, boxes, frame_of, ids, exists = [], [], [], [], []
        for fr in range(n_frames):
            present = [k for k in range(5) if rng.uniform() < 0.6]
            rng.shuffle(present)
            for k in present:
                centres[k] += rng.normal(0, 6, 2)
                noise = rng.normal(0, 1, D).astype(np.float32)
                noise /= np.linalg.norm(noise)
                f = protos[k] + np.float32(rng.choice([0.3, 0.6, 0.85, 1.2])) * noise
                if kind == "MOBILE_FACENET":
                    f = (f / np.linalg.norm(f)).astype(np.float32)
                else:
                    f = (f * np.float32(rng.uniform(0.5, 3.0))).astype(np.float32)
                half = rng.integers(30, 80)
                b = (int(centres[k][0] - half), int(centres[k][1] - half), int(centres[k][0] + half), int(centres[k][1] + half))
                with contextlib.redirect_stdout(io.StringIO()):
                    ok, fid, _, _ = net.check_if_face_exists(f, b)
                    if not ok:
                        fid = net.max_faceid + 1
                        net.add_face(f, b, None, None)
                feats.append(f); boxes.append(b); frame_of.append(fr); ids.append(fid); exists.append(ok)
        tag = "l2" if kind == "MOBILE_FACENET" else "cos"
        out[f"{tag}_feats"] = np.stack(feats)
        out[f"{tag}_boxes"] = np.asarray(boxes, np.int32)
        out[f"{tag}_frame"] = np.asarray(frame_of, np.int32)
        out[f"{tag}_ids"] = np.asarray(ids, np.int32)
        out[f"{tag}_exists"] = np.asarray(exists, bool)
        out[f"{tag}_final_feats"] = np.stack([e[1] for e in net.face_feat_bbox_age_gender_list])
        out[f"{tag}_final_boxes"] = np.asarray([e[2] for e in net.face_feat_bbox_age_gender_list], np.int32)
    # IoU helper itself (fde/modules/utils/image.py:124-143)
    from modules.utils.image import calculate_bbox_iou
    rng = np.random.default_rng(9)
    pairs = rng.integers(0, 200, (64, 2, 2))
    b1 = np.concatenate([pairs[:, 0], pairs[:, 0] + rng.integers(1, 120, (64, 2))], 1)
    b2 = np.concatenate([pairs[:, 1], pairs[:, 1] + rng.integers(1, 120, (64, 2))], 1)
    out["iou_b1"], out["iou_b2"] = b1.astype(np.int32), b2.astype(np.int32)
    out["iou"] = np.asarray([calculate_bbox_iou(tuple(int(v) for v in a), tuple(int(v) for v in b)) for a, b in zip(b1, b2)],
                            np.float64)
    save("tracker", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    ns = rh.import_reference()
    which = sys.argv[1:] or ["blazeface", "mobilefacenet", "utils", "similarity", "yolo", "tracker"]
    if "blazeface" in which:
        gen_blazeface(ns)
    if "mobilefacenet" in which:
        gen_mobilefacenet(ns)
    if "utils" in which:
        gen_utils(ns)
    if "similarity" in which:
        gen_similarity()
    if "yolo" in which:
        gen_yolo()
    if "tracker" in which:
        gen_tracker()
    if "yolo_blocks" in which:      # added in round 2; not part of the default list so the round-1 files stay untouched
        gen_yolo_blocks()
    if "jpeg" in which:             # round 4
        gen_jpeg()

"""GPU tests of the reference-named entry points (CLI surface + plugin API) on synthetic weights / images."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from face_detection_and_recognition_amd import workload as W
from face_detection_and_recognition_amd.modules.blazeface.blazeface import generate_anchors
from face_detection_and_recognition_amd.modules.mobile_facenet.mobile_facenet import MobileFaceNet
from face_detection_and_recognition_amd.modules.utils.inference import get_dets_bboxes_confs_lmarks_areas
from face_detection_and_recognition_amd.synth import synth_state_dict
from conftest import rel_err
from oracle import blazeface_ref, image_ref, mobilefacenet_ref, similarity_ref

pytestmark = pytest.mark.gpu


def _save_png_as_jpg_free(path, arr):
    Image.fromarray(arr[..., ::-1]).save(path, quality=95)       # arr is BGR


def test_detect_face_blazeface_cli_matches_oracle(dev, tmp_path):
    from face_detection_and_recognition_amd import detect_face_blazeface
    from face_detection_and_recognition_amd.modules.utils.inference import load_image
    frames = W.make_frames(8, dev, seed=3)
    det = W.build_detector(dev, frames, cand_per_frame=48)
    wpath = str(tmp_path / "blazefaceback.pth")
    torch.save({k: v.cpu() for k, v in det.net.state_dict().items()}, wpath)
    np.save(str(tmp_path / "anchors.npy"), generate_anchors(True))
    ipath = str(tmp_path / "frame.jpg")
    _save_png_as_jpg_free(ipath, frames[0].cpu().numpy())
    post = detect_face_blazeface.main(["-i", ipath, "--md", wpath, "--mt", "back", "--dt", "0.7", "--at", "0.12",
                                       "-d", "hip:0"])
    # oracle on the decoded jpg (same decoder), reference flow: letterbox -> net -> decode -> wNMS -> B7
    img = load_image(ipath)
    sd = {k: v.cpu() for k, v in det.net.state_dict().items()}
    lb = image_ref.pad_resize_image(img, (256, 256))[..., ::-1].copy()
    faces, _ = blazeface_ref.predict_on_batch(sd, torch.from_numpy(lb).permute(2, 0, 1).unsqueeze(0),
                                              torch.from_numpy(generate_anchors(True)), True)
    d = faces[0].numpy()[:, [1, 0, 3, 2] + list(range(4, 17))]
    ref = image_ref.dets_to_boxes(d.copy(), (img.shape[1], img.shape[0]), (256, 256), 0.7, 0.12)
    assert len(post.boxes) == len(ref["boxes"]) > 0
    assert np.abs(post.boxes - ref["boxes"]).max() <= 1.0          # rounded pixel coordinates
    np.testing.assert_allclose(post.bbox_confs, ref["bbox_confs"], atol=1e-4)
    with pytest.raises(NotImplementedError):
        detect_face_blazeface.main(["-i", ipath, "--md", str(tmp_path / "x.tflite"), "-d", "hip"])
    with pytest.raises(FileNotFoundError):
        detect_face_blazeface.main(["-i", str(tmp_path / "missing.jpg"), "--md", wpath, "-d", "hip"])


def test_detect_face_blazeface_cli_front_camera(dev, tmp_path):
    """BASELINE configs[0]: detect_face_blazeface.py --mt front on one 576x1024 image (128x128 front-camera network,
    threshold 0.75), against the oracle's reference flow on the decoded file."""
    from face_detection_and_recognition_amd import detect_face_blazeface
    from face_detection_and_recognition_amd.modules.blazeface.blazeface import BlazeFace
    from face_detection_and_recognition_amd.modules.utils.inference import load_image
    frames = W.make_frames(4, dev, seed=21)
    net = BlazeFace(back_model=False)
    sd = synth_state_dict(net.state_dict(), 111, residual_gain=0.5)
    for name in ("regressor_8", "regressor_16"):            # well-formed boxes of ~40 input pixels
        sd[name + ".weight"] = sd[name + ".weight"] * 0.05
        b = sd[name + ".bias"] * 0.0
        b.view(-1, 16)[:, 2:4] = 40.0
        sd[name + ".bias"] = b
    for name in ("classifier_8", "classifier_16"):
        sd[name + ".weight"] = sd[name + ".weight"].abs() * 0.5
        sd[name + ".bias"] = sd[name + ".bias"] * 0.0
    # shift the classifier bias so ~40 of the 896 anchors reach the front model's 0.75 threshold (oracle forward)
    ipath = str(tmp_path / "frame.jpg")
    _save_png_as_jpg_free(ipath, frames[0].cpu().numpy())
    img = load_image(ipath)
    assert img.shape == (576, 1024, 3)
    lb = image_ref.pad_resize_image(img, (128, 128))[..., ::-1].copy()
    x_u8 = torch.from_numpy(lb).permute(2, 0, 1).unsqueeze(0)
    anchors = torch.from_numpy(generate_anchors(False))
    with torch.no_grad():
        _, c = blazeface_ref.forward(sd, x_u8.float() / 127.5 - 1.0, False)
    kth = torch.sort(c.flatten(), descending=True)[0][40]
    delta = float(np.log(0.75 / 0.25) - kth) + 0.01
    for name in ("classifier_8", "classifier_16"):
        sd[name + ".bias"] = sd[name + ".bias"] + delta
    wpath = str(tmp_path / "blazeface.pth")
    torch.save(sd, wpath)
    np.save(str(tmp_path / "anchors.npy"), generate_anchors(False))
    post = detect_face_blazeface.main(["-i", ipath, "--md", wpath, "--mt", "front", "--dt", "0.7", "--at", "0.12",
                                       "-d", "hip:0"])
    faces, _ = blazeface_ref.predict_on_batch(sd, x_u8, anchors, False)
    d = faces[0].numpy()[:, [1, 0, 3, 2] + list(range(4, 17))]
    ref = image_ref.dets_to_boxes(d.copy(), (img.shape[1], img.shape[0]), (128, 128), 0.7, 0.12)
    assert len(post.boxes) == len(ref["boxes"]) > 0
    assert np.abs(post.boxes - ref["boxes"]).max() <= 1.0          # rounded pixel coordinates
    np.testing.assert_allclose(post.bbox_confs, ref["bbox_confs"], atol=1e-4)
    np.testing.assert_allclose(post.bbox_areas, ref["bbox_areas"], rtol=1e-4)


def test_blank_frames_give_empty_results_everywhere(dev):
    """The reference's blank-image tests (fde/tests/image_tests/test_blazeface.py:9-24, test_yolov5_face.py:9-41) expect
    zero detections with the right empty shapes; here the detectors are made silent through their head biases (no
    trained weights offline) and the whole batched path must cope with zero faces: empty (0, 17) / (0, 5) arrays from
    the plugin API, a step with n_faces = 0, no embedder launch, no filter output -- and recover on the next batch."""
    from face_detection_and_recognition_amd.modules.yolov5_face.model import YOLOV5FaceModel
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    blank = np.full((3, 576, 1024, 3), 117, np.uint8)
    frames = W.make_frames(3, dev, seed=41)
    det = W.build_detector(dev, W.make_frames(8, dev, seed=42), cand_per_frame=48)
    emb = W.build_embedder(dev)
    pipe = FacePipeline(det, emb, W.make_reference(32, dev), tau=0.0)
    n_before = pipe.step(frames)["n_faces"]
    assert n_before > 0
    with torch.no_grad():                                   # silence the detector
        det.net.classifier_8.bias -= 50.0
        det.net.classifier_16.bias -= 50.0
    det.net._plans.clear()
    out = pipe.step(torch.from_numpy(blank).to(dev))
    assert out["n_faces"] == 0 and out["emb"].shape == (0, 512) and out["info"].shape[0] == 0 and "keep" not in out
    d = det(blank[0])
    assert isinstance(d, np.ndarray) and d.shape == (0, 17)
    assert [x.shape for x in det.predict_batch(blank)] == [(0, 17)] * 3
    with torch.no_grad():
        det.net.classifier_8.bias += 50.0
        det.net.classifier_16.bias += 50.0
    det.net._plans.clear()
    assert pipe.step(frames)["n_faces"] == n_before          # the pipeline recovers after an empty step
    ydet = W.build_yolo_detector(dev, W.make_frames(4, dev, seed=43), "yolov5n-0.5", cand_per_frame=20)
    with torch.no_grad():
        for conv in ydet.net.model[-1].m:
            conv.bias.view(3, 16)[:, 4] -= 60.0
    ydet.net._plans.clear()
    d5 = ydet(blank[0])
    assert isinstance(d5, np.ndarray) and d5.shape == (0, 5)
    ypipe = FacePipeline(ydet, emb, None)
    assert ypipe.step(torch.from_numpy(blank).to(dev))["n_faces"] == 0


@pytest.mark.parametrize("two_streams", [False, True])
def test_pipeline_step_overlapped_equals_step(dev, two_streams):
    """FacePipeline.step_overlapped (the detector of batch k + 1 enqueued before the host reads batch k's face count; with
    two_streams: embed + filter of batch k on a side stream beside it) returns, one call late, exactly what step() returns
    for the same batches; flush() hands out the last one.  Five batches, so that the side stream really runs beside the
    next detector pass and the embedder arena is reused while results of earlier batches are still held."""
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    det = W.build_detector(dev, W.make_frames(8, dev, seed=8), cand_per_frame=48)
    emb = W.build_embedder(dev)
    ref = W.make_reference(300, dev)
    pipe = FacePipeline(det, emb, ref, tau=0.1, two_streams=two_streams)
    batches = [W.make_frames(6, dev, seed=40 + i) for i in range(5)]
    want = [pipe.step(b, beside=two_streams) for b in batches]     # the detector plan step_overlapped will use (blazeface.py co_scheduled)
    got = [pipe.step_overlapped(b) for b in batches]
    assert got[0] is None
    got = got[1:] + [pipe.flush()]
    assert pipe.flush() is None
    torch.cuda.synchronize()
    for a, b in zip(want, got):
        assert ("done" in b) == two_streams
        assert a["n_faces"] == b["n_faces"] > 0
        for k in ("emb", "info", "items", "best", "arg", "keep"):
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("two_streams", [False, True])
def test_embedder_split_tail_is_bit_identical(dev, two_streams):
    """FacePipeline.embed runs a batch a little above a multiple of 512 crops as that multiple + the remainder on a side
    stream (pipeline._embed_split: no nearly empty last round of workgroups in the Depth_Wise kernels).  Same rows, bit for
    bit, as the one-run form, over several steps (the small plan's arena is reused while earlier results are held), on a
    full bench batch (256 frames, ~523 faces -> 512 + 16)."""
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    det = W.build_detector(dev, W.make_frames(64, dev, seed=999))
    emb = W.build_embedder(dev)
    ref = W.make_reference(1000, dev)
    batches = [W.make_frames(256, dev, seed=1234 + i) for i in range(3)]
    outs = {}
    for split in (False, True):
        pipe = FacePipeline(det, emb, ref, tau=0.3, two_streams=two_streams, split_tail=split)
        got = [pipe.step_overlapped(b) for b in batches] + [pipe.flush()]
        torch.cuda.synchronize()
        outs[split] = got[1:]
    counts = [a["n_faces"] for a in outs[False]]
    assert any(512 < n <= 512 + FacePipeline.TAIL_MAX for n in counts), counts         # at least one batch that splits
    for a, b in zip(outs[False], outs[True]):
        assert a["n_faces"] == b["n_faces"]
        for k in ("emb", "info", "items", "best", "arg", "keep"):
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("n", [1, 511, 512, 513, 520, 607, 608, 609, 1030])
def test_embed_split_boundaries(dev, n):
    """FacePipeline.embed on fabricated crop items around the split's boundaries: exactly a multiple of 512 crops, one crop more
    (a remainder padded to 8), the largest remainder that is split off (96) and one more (not split), a second multiple of 512
    + 6; the rows equal the one-run form bit for bit and the padding crops never leak into them."""
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    emb = W.build_embedder(dev)
    frames = W.make_frames(4, dev, seed=77)
    rng = np.random.default_rng(n)
    items = np.zeros((n + 16, 9), np.int32)
    for k in range(n + 16):
        w, h = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        items[k] = [k % 4, int(rng.integers(0, 1024 - w)), int(rng.integers(0, 576 - h)), w, h, 0, 0, 112, 112]
    it = torch.from_numpy(items).to(dev)
    outs = []
    for split in (False, True):
        pipe = FacePipeline(None, emb, None, split_tail=split)
        e = pipe.embed(frames, it, n).clone()
        torch.cuda.synchronize()
        outs.append(e)
        if split:
            n_pad = (n + 7) // 8 * 8
            main = n_pad // 512 * 512
            did_split = main and 0 < n_pad - main <= FacePipeline.TAIL_MAX and main < n
            assert pipe.emb_plan.n_run == (main if did_split else n_pad), (n, pipe.emb_plan.n_run, main)
    assert outs[0].shape == (n, 512) and torch.equal(outs[0], outs[1])
    np.testing.assert_allclose(torch.linalg.norm(outs[1], dim=1).cpu().numpy(), 1.0, atol=1e-5)


def test_filter_faces_using_reference_cli(dev, tmp_path):
    from face_detection_and_recognition_amd.similar_face_filtering import filter_faces_using_reference as F
    assert F._fix_path_for_globbing("data/") == "data/*" and F._fix_path_for_globbing("data") == "data/*"
    rng = np.random.default_rng(0)
    net = MobileFaceNet(512)
    net.load_state_dict(synth_state_dict(net.state_dict(), 300))
    wpath = str(tmp_path / "mfn.pth")
    torch.save(net.state_dict(), wpath)
    base = {c: rng.integers(0, 256, (64, 64, 3), dtype=np.uint8) for c in ("class_a", "class_b")}
    for root, n in (("ref", 6), ("unf", 9)):
        for c in base:
            os.makedirs(tmp_path / root / c)
            for i in range(n):
                noise = rng.integers(-40, 40, base[c].shape) if (root == "unf" and i % 3 == 2) else \
                    rng.integers(-4, 4, base[c].shape)
                img = np.clip(base[c].astype(int) + noise, 0, 255).astype(np.uint8)
                Image.fromarray(img).save(tmp_path / root / c / f"{c}_{i}.jpg", quality=95)
    assert F.get_class_name_list(str(tmp_path / "ref")) == ["class_a", "class_b"]
    F.main(["--ud", str(tmp_path / "unf"), "--rd", str(tmp_path / "ref"), "--td", str(tmp_path / "out"), "-m", wpath,
            "-b", "4", "-r", "32", "-d", "hip:0"])
    for c in base:
        clean = os.listdir(tmp_path / "out" / "clean" / c)
        unclean = os.listdir(tmp_path / "out" / "unclean" / c)
        assert len(clean) + len(unclean) == 9
    # the filter decision equals the reference arithmetic on the same embeddings (oracle network + numpy filter)
    netd = net.to(dev)
    paths = sorted(str(p) for p in (tmp_path / "unf" / "class_a").glob("*.jpg"))
    refs = [str(p) for p in (tmp_path / "ref" / "class_a").glob("*.jpg")]
    e_hip = F.embed_images(netd, paths, 4).cpu().numpy()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}

    def emb_ref(p):
        img = F.read_image_bgr(p)
        face = image_ref.mfn_lut()[image_ref.resize_bilinear_u8(img, (112, 112))]
        with torch.no_grad():
            return mobilefacenet_ref.forward(sd, torch.from_numpy(np.ascontiguousarray(face.transpose(2, 0, 1))).unsqueeze(0))[0].numpy()
    e_ref = np.stack([emb_ref(p) for p in paths])
    assert np.abs(e_hip - e_ref).max() < 1e-4
    mean, thres = similarity_ref.ref_mean_and_thres(np.stack([emb_ref(p) for p in refs]))
    dist, keep = similarity_ref.l2_filter(e_ref, mean, thres)
    got_clean = set(os.listdir(tmp_path / "out" / "clean" / "class_a"))
    margin = np.abs(dist - thres) > 1e-3
    for p, k, mg in zip(paths, keep, margin):
        if mg:
            assert (os.path.basename(p) in got_clean) == bool(k)
    # --preprocess tf_standardize: the reference filter's own read_and_preprocess_img (:60-68) in front of the network
    # (at its 112x112 input), end to end through the CLI and against the oracle's restatement of that preprocess
    F.main(["--ud", str(tmp_path / "unf"), "--rd", str(tmp_path / "ref"), "--td", str(tmp_path / "out_tf"), "-m", wpath,
            "-b", "4", "--preprocess", "tf_standardize", "-d", "hip:0"])
    assert sum(len(os.listdir(tmp_path / "out_tf" / d / "class_a")) for d in ("clean", "unclean")) == 9
    e_tf = F.embed_images(netd, paths[:3], 4, preprocess="tf_standardize").cpu().numpy()

    def emb_ref_tf(p):
        rgb = np.ascontiguousarray(F.read_image_bgr(p)[..., ::-1])
        x = image_ref.read_and_preprocess_rgb(rgb, (112, 112)).astype(np.float32)
        with torch.no_grad():
            return mobilefacenet_ref.forward(sd, torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1))).unsqueeze(0))[0].numpy()
    assert np.abs(e_tf - np.stack([emb_ref_tf(p) for p in paths[:3]])).max() < 1e-4
    assert np.abs(e_tf - e_hip[:3]).max() > 1e-3          # it IS a different preprocess
    with pytest.raises(ValueError):
        F.embed_images(netd, paths[:1], 1, preprocess="nope")
    with pytest.raises(Exception):
        os.makedirs(tmp_path / "ref" / "class_c")
        F.main(["--ud", str(tmp_path / "unf"), "--rd", str(tmp_path / "ref"), "--td", str(tmp_path / "o2"), "-m", wpath])


def test_yolo_entry_point_and_loader(dev, tmp_path):
    from face_detection_and_recognition_amd import detect_face_yolov5_face as E
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import Model
    m = Model("yolov5n")
    m.load_state_dict(synth_state_dict(m.state_dict(), 5))
    wpath = str(tmp_path / "yolov5n-face.pt")
    torch.save(m.state_dict(), wpath)
    model = E.load_model(wpath, 0.4, 0.0, (640, 640), "hip:0")
    frame = np.random.default_rng(1).integers(0, 256, (360, 640, 3), dtype=np.uint8)
    dets = model(frame)
    assert dets.ndim == 2 and dets.shape[1] == 5
    with pytest.raises(NotImplementedError):
        E.load_model(str(tmp_path / "x.onnx"), 0.4, 0.0, (640, 640), "hip")
    bad = str(tmp_path / "yolov5s-face.pt")
    torch.save({"not": "a state dict"}, bad)
    with pytest.raises(Exception):
        E.load_model(bad, 0.4, 0.0, (640, 640), "hip")


def test_extract_faces_batched_matches_oracle_and_format(dev, tmp_path):
    """detect -> crop(+offsets) -> embed per frame, saved in the reference's .npy dict format."""
    from face_detection_and_recognition_amd.face_extraction import extract_faces_from_dataset as X
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    frames = W.make_frames(6, dev, seed=11)
    det = W.build_detector(dev, W.make_frames(8, dev, seed=12), cand_per_frame=48)
    emb = W.build_embedder(dev)
    pipe = FacePipeline(det, emb, None)
    recs = X.extract_face_feat_conf_area_list(pipe, frames, frame_nums=list(range(10, 16)))
    assert len(recs) == 6 and recs[0].frame_num == 10
    # oracle, reference flow per frame (blazeface path + crop + embed)
    sd_det = {k: v.cpu() for k, v in det.net.state_dict().items()}
    sd_emb = {k: v.cpu() for k, v in emb.state_dict().items()}
    fr = frames.cpu().numpy()
    for i, f in enumerate(fr):
        lb = image_ref.pad_resize_image(f, (256, 256))[..., ::-1].copy()
        faces, _ = blazeface_ref.predict_on_batch(sd_det, torch.from_numpy(lb).permute(2, 0, 1).unsqueeze(0),
                                                  det.net.anchors.cpu(), True)
        d = faces[0].numpy()
        if len(d) == 0:
            assert len(recs[i].confs) == 0
            continue
        d = d[:, [1, 0, 3, 2] + list(range(4, 17))]
        post = image_ref.dets_to_boxes(d.copy(), (f.shape[1], f.shape[0]), (256, 256), det.det_thres, det.bbox_area_thres)
        assert len(post["boxes"]) == len(recs[i].confs)
        np.testing.assert_array_equal(recs[i].boxes, post["boxes"].astype(np.float32))
        np.testing.assert_allclose(recs[i].areas, post["bbox_areas"], rtol=1e-6)
        for k, box in enumerate(post["boxes"]):
            crop, _ = image_ref.crop_face(f, box)
            face = image_ref.mfn_lut()[image_ref.resize_bilinear_u8(crop, (112, 112))]
            with torch.no_grad():
                e = mobilefacenet_ref.forward(sd_emb, torch.from_numpy(np.ascontiguousarray(face.transpose(2, 0, 1))).unsqueeze(0))[0].numpy()
            assert np.abs(recs[i].feats[k] - e).max() < 1e-4
    total = X.save_extracted_faces(recs, "vid0", "person_a", str(tmp_path / "feats"), 512, {"person_a": 7})
    a = np.load(tmp_path / "feats" / "vid0.npy", allow_pickle=True).item()     # file written by this test
    assert total == sum(len(r.confs) for r in recs)
    assert a["media_id"] == "vid0" and a["label"] == 7 and len(a["frames_info"]) == 6
    assert a["feature"].shape == (X.MAX_N_FRAME_FROM_VID * X.MAX_N_FACES_PER_FRAME * 512,) and a["feature"].dtype == np.float32


def test_yolov5s_to_mobilefacenet_pipeline_matches_oracle(dev):
    """BASELINE configs[3]: YOLOv5s-face detect -> crops -> Mobile-FaceNet, the reference's own composition
    (fde/face_extraction/extract_faces_from_dataset.py:270-307 with bbox_conf_area_func = get_bboxes_confs_areas,
    fde/modules/yolov5_face/onnx/onnx_utils.py:313-340) through FacePipeline on 576x1024 frames.
      * forward: the HIP decoded predictions against the oracle network, within tolerance;
      * NMS kept rows, filtered boxes and the fmt = 1 crop rectangles: exact, with the oracle post-processing run on
        the SAME decoded predictions (so the comparison is bit-for-bit, not blurred by the forward tolerance);
      * embeddings: <= 1e-4 against the oracle Mobile-FaceNet on the oracle's crops."""
    from face_detection_and_recognition_amd.modules.yolov5_face import preprocess_batch
    from face_detection_and_recognition_amd.modules.yolov5_face.yolo import SPECS
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    from oracle import yolo_ref
    B = 4
    frames = W.make_frames(B, dev, seed=31)
    det = W.build_yolo_detector(dev, W.make_frames(4, dev, seed=32), "yolov5s", cand_per_frame=80)
    emb = W.build_embedder(dev)
    ref = W.make_reference(64, dev)
    pipe = FacePipeline(det, emb, ref, tau=0.0, max_faces_per_frame=64)
    out = pipe.step(frames)
    torch.cuda.synchronize()
    n = out["n_faces"]
    assert n >= B, "workload calibration found too few faces"
    info, items, got_emb = out["info"].cpu().numpy(), out["items"].cpu().numpy(), out["emb"].cpu().numpy()

    # the decoded predictions the device NMS consumed (same plan, same input)
    plan = preprocess_batch(det.net, frames, det.input_size)
    z = det.net.run_plan(plan).clone().cpu()
    fr = frames.cpu().numpy()
    sd_det = {k: v.detach().cpu() for k, v in det.net.state_dict().items()}
    sd_emb = {k: v.detach().cpu() for k, v in emb.state_dict().items()}
    lb = image_ref.pad_resize_image(fr[0][..., ::-1], (640, 640))
    x0 = torch.from_numpy(image_ref.yolo_lut()[lb]).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        z_ref, _ = yolo_ref.forward(SPECS["yolov5s"], sd_det, x0)
    assert rel_err(z[0].numpy(), z_ref[0].numpy()) < 1e-4

    kept, _ = yolo_ref.non_max_suppression_face(z.numpy(), 0.4, 0.5)
    row = 0
    embs = []
    for i in range(B):
        boxes, confs, areas = yolo_ref.get_bboxes_confs_areas(kept[i].numpy(), det.det_thres, det.bbox_area_thres,
                                                              (1024, 576), (640, 640))
        areas = areas[areas > det.bbox_area_thres]
        k = len(boxes)
        np.testing.assert_array_equal(info[row:row + k, 0], i)
        np.testing.assert_array_equal(info[row:row + k, 1:5], boxes)               # kept + filtered boxes, exact
        np.testing.assert_array_equal(info[row:row + k, 5], confs)
        np.testing.assert_array_equal(info[row:row + k, 6], areas)                 # percent, as the reference returns
        for j, box in enumerate(boxes):
            crop, (x, y, xw, yh) = image_ref.crop_face(fr[i], box)
            assert tuple(items[row + j, :5]) == (i, x, y, xw - x, yh - y)          # fmt = 1 crop rectangle, exact
            face = image_ref.mfn_lut()[image_ref.resize_bilinear_u8(crop, (112, 112))]
            xin = torch.from_numpy(np.ascontiguousarray(face.transpose(2, 0, 1))).unsqueeze(0)
            with torch.no_grad():
                embs.append(mobilefacenet_ref.forward(sd_emb, xin)[0].numpy())
        row += k
    assert row == n
    assert np.abs(got_emb - np.stack(embs)).max() < 1e-4
    # plugin API on the same detector: (k, 5) rows normalised to the input size (yolov5_face/model.py:23-37)
    d5 = det(fr[0])
    assert d5.shape == (len(kept[0]), 5)
    np.testing.assert_allclose(d5[:, :4] * 640, kept[0].numpy()[:, :4], rtol=0, atol=1e-3)
    # results of one step survive the next one (outputs used to alias the plan arena)
    e0 = out["emb"].clone()
    pipe.step(W.make_frames(B, dev, seed=33))
    torch.cuda.synchronize()
    assert torch.equal(e0, out["emb"])


@pytest.mark.parametrize("tag,kind,batched", [("l2", "MOBILE_FACENET", True), ("cos", "FACE_REID_MNV3", True),
                                               ("l2", "MOBILE_FACENET", False)])
def test_face_tracker_matches_reference_golden(dev, tag, kind, batched):
    """csrc/tracker.hip through FaceTracker (the reference's Net.check_if_face_exists / add_face rules,
    extract_and_label_faces_from_dataset.py:101-121) against ids produced by the reference itself: one launch per
    frame (batched) or the per-face API; final gallery state must equal the reference's list."""
    from conftest import golden
    from face_detection_and_recognition_amd.face_extraction.face_tracker import FaceTracker
    g = golden("tracker")
    feats, boxes, frame = g[tag + "_feats"], g[tag + "_boxes"], g[tag + "_frame"]
    tr = FaceTracker(kind, feat_dim=feats.shape[1], device=dev, max_faces=32)
    ids, exists = [], []
    if batched:
        for fr in np.unique(frame):
            sel = np.nonzero(frame == fr)[0]
            i, e = tr.track(torch.from_numpy(feats[sel]), torch.from_numpy(boxes[sel]))
            ids += i.cpu().tolist()
            exists += [bool(v) for v in e.cpu().tolist()]
    else:
        for f, b in zip(feats, boxes):
            ok, fid, _, _ = tr.check_if_face_exists(f, tuple(int(v) for v in b))
            if not ok:
                fid = tr.get_num_unique_faces()
                tr.add_face(f, b, 30, "M")
            ids.append(fid)
            exists.append(ok)
    np.testing.assert_array_equal(np.asarray(ids, np.int32), g[tag + "_ids"])
    np.testing.assert_array_equal(np.asarray(exists, bool), g[tag + "_exists"])
    n = tr.get_num_unique_faces()
    assert n == len(g[tag + "_final_boxes"])
    np.testing.assert_array_equal(tr.feats[:n].cpu().numpy(), g[tag + "_final_feats"])
    np.testing.assert_array_equal(tr.bboxes[:n].cpu().numpy(), g[tag + "_final_boxes"])


def test_face_tracker_capacity_and_empty(dev):
    from face_detection_and_recognition_amd.face_extraction.face_tracker import FaceTracker
    tr = FaceTracker("MOBILE_FACENET", feat_dim=8, device=dev, max_faces=2)
    i, e = tr.track(torch.zeros((0, 8)), torch.zeros((0, 4), dtype=torch.int32))
    assert i.numel() == 0 and tr.get_num_unique_faces() == 0
    f = torch.eye(8)[:3] * 10          # three mutually distant features, disjoint boxes
    b = torch.tensor([[0, 0, 10, 10], [100, 100, 120, 120], [300, 300, 320, 320]], dtype=torch.int32)
    i, e = tr.track(f, b)
    assert i.cpu().tolist() == [1, 2, 0] and e.cpu().tolist() == [0, 0, 0]     # third face: gallery full -> id 0
    tr.clear_faces()
    assert tr.get_num_unique_faces() == 0
    with pytest.raises(ValueError):
        FaceTracker("MOBILE_FACENET", device=dev, use_bbox_iou_to_track_face=False)


def _triton_request(rng, n_faces, h=384, w=640):
    """Three raw head tensors (1, 3, ny, nx, 16) with n_faces strong, well separated objects, and an RGB image."""
    heads = []
    for stride in (8, 16, 32):
        t = rng.normal(0, 0.5, (1, 3, h // stride, w // stride, 16)).astype(np.float32)
        t[..., 4] = -8.0                     # objectness logit: nothing fires ...
        heads.append(t)
    for k in range(n_faces):                 # ... except these cells of the stride-16 head (anchor 1: 43x55 px)
        gy, gx = 4 + 5 * (k // 4), 4 + 9 * (k % 4)
        heads[1][0, 1, gy, gx, 4] = 3.0 + 0.3 * k
        heads[1][0, 1, gy, gx, 0:4] = rng.normal(0, 0.3, 4) + np.array([0, 0, 0.8, 0.8], np.float32)
    img = rng.uniform(0, 1, (1, 3, h, w)).astype(np.float32)
    return dict(stride_8_out=heads[0], stride_16_out=heads[1], stride_32_out=heads[2], images=img,
                face_det_thres=np.asarray([0.7], np.float32), face_bbox_area_thres=np.asarray([0.1], np.float32))


@pytest.mark.parametrize("n_faces", [5, 0])
def test_triton_postprocess_contract_matches_oracle(dev, n_faces):
    """modules/face_detection_trt_server/yolov5_face_postprocess.py (decode + w-NMS + float crop/resize kernels) against
    oracle/triton_postprocess_ref.py, the restatement of the Triton python model (model.py:32-113): same boxes and
    confidences, faces within 1e-4, and the zero-filled (1,3,112,112) / [[0,0,0,0]] / [[0.]] reply when nothing fires."""
    from face_detection_and_recognition_amd.modules.face_detection_trt_server.yolov5_face_postprocess import \
        YOLOv5FacePostprocess
    from oracle import triton_postprocess_ref as ref
    rng = np.random.default_rng(31 + n_faces)
    req = _triton_request(rng, n_faces)
    out = YOLOv5FacePostprocess(dev).execute([req])[0]
    faces, boxes, confs = ref.execute(req["stride_8_out"], req["stride_16_out"], req["stride_32_out"], req["images"],
                                      req["face_det_thres"][0], req["face_bbox_area_thres"][0])
    assert out["face_detector_faces"].dtype == np.float32 and out["face_detector_bboxes"].dtype == np.int32
    assert out["face_detector_bboxes"].shape == boxes.shape and out["face_detector_faces"].shape == faces.shape
    np.testing.assert_array_equal(out["face_detector_bboxes"], boxes)
    np.testing.assert_allclose(out["face_detector_confs"], confs, rtol=0, atol=2e-6)
    np.testing.assert_allclose(out["face_detector_faces"], faces, rtol=0, atol=1e-4)
    if n_faces:
        assert len(boxes) == n_faces and faces.shape[1:] == (3, 112, 112)
    else:
        assert faces.shape == (1, 3, 112, 112) and boxes.tolist() == [[0, 0, 0, 0]]


def test_tf_style_preprocess_matches_oracle_and_reference_test_property(dev):
    """fp_resize_standardize (filter_faces_using_reference.py:60-68) against oracle/image_ref.py, and the
    property the reference's own test pins (sff/tests/base/test_similar_faces_filter.py:19-27): at the native size the
    result equals (img - mean)/max(std, 1/sqrt(N)) of the uint8 image within 1e-4."""
    from face_detection_and_recognition_amd.similar_face_filtering.filter_faces_using_reference import \
        preprocess_tf_standardize
    rng = np.random.default_rng(8)
    frames = rng.integers(0, 256, (3, 97, 131, 3), dtype=np.uint8)
    got = preprocess_tf_standardize(torch.from_numpy(frames).to(dev), (160, 160)).cpu().numpy()
    for i in range(3):
        np.testing.assert_allclose(got[i], image_ref.read_and_preprocess_rgb(frames[i], (160, 160)), rtol=0, atol=1e-4)  # fma in the lerp
    same = rng.integers(0, 256, (2, 160, 160, 3), dtype=np.uint8)
    got = preprocess_tf_standardize(torch.from_numpy(same).to(dev), (160, 160)).cpu().numpy()
    for i in range(2):
        img = same[i]
        n = img.size
        ref = (img - np.mean(img)) / max(np.std(img), 1 / (n ** 0.5))
        assert np.allclose(got[i], ref, atol=1e-4)
    const = np.full((1, 40, 40, 3), 77, np.uint8)                         # std = 0 -> divided by 1/sqrt(N), all zeros
    assert np.abs(preprocess_tf_standardize(torch.from_numpy(const).to(dev), (160, 160)).cpu().numpy()).max() < 1e-4


def test_bench_gpus_2_self_launch_on_one_gpu(dev):
    """`python bench.py --gpus 2`, end to end with the REAL step: the launcher starts two ranks, both run the whole
    detect -> embed -> filter step and the overlapped cross-rank exchange (StepExchange) on this box's one GPU
    (BENCH_SAME_DEVICE=1; gloo carries the collectives because two RCCL ranks cannot share a device), and the single line
    on stdout reports n_gpus = 2 = the process group's own world size, with both ranks' face counts.  (The RCCL form of the
    same code path is rehearsed with BENCH_FORCE_DIST=1; the 8-GPU run is the driver's.)"""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(BENCH_DIST_BACKEND="gloo", BENCH_SAME_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--no-other-configs"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["launch"]["world_size"] == 2 and rec["launch"]["requested_gpus"] == 2
    per_rank = rec["launch"]["units_per_rank"]
    assert len(per_rank) == 2 and all(4 * 256 <= n <= 4 * 256 * 4 for n in per_rank)         # ~2 faces per frame, 4 steps each
    assert abs(rec["value"] * rec["ms_per_step"] * 4e-3 - sum(per_rank)) <= 1.0                # whole-job faces / max-rank time
    assert rec["config"]["frames_per_step_per_gpu"] == 256 and "gloo" in rec["launch"]["backend"]

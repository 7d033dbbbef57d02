"""Measures the other BASELINE.json configs on one GPU (they are parity-test cases, not the headline bench line):
  [2] YOLOv5n-face 640x640 batch 256 + batched NMS
  [3] YOLOv5s-face detect -> Mobile-FaceNet 112x112, 1024 crops
  [4] cosine filter 1M gallery x 10k reference x 512-d (single-GPU share and the full problem)
Seeded synthetic weights/inputs; prints one JSON line per config."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_detection_and_recognition_amd import similarity as S  # noqa: E402
from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd.modules.yolov5_face import nms_face_device, preprocess_batch  # noqa: E402


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# SURVEY 8(d) / BASELINE.md section 3: per-GPU rooflines of the op-granular model (img/s, crops/s, TFLOP/s)
ROOF_YOLO_IMG_S = {"yolov5n": 28.5e3, "yolov5s": 19.2e3}
ROOF_MFN_CROPS_S = 310e3
ROOF_FP32_MFMA_TF = 157.3


def yolo(name, dev, B=256, cand_per_frame=80):
    """BASELINE configs[2]: letterbox + forward + Detect decode + batched NMS with ~cand_per_frame candidates per image
    (objectness calibrated off the clock, workload.build_yolo_detector)."""
    det = W.build_yolo_detector(dev, W.make_frames(16, dev, seed=6), name, cand_per_frame=cand_per_frame)
    m = det.net
    frames = W.make_frames(B, dev, seed=5)
    plan = preprocess_batch(m, frames, (640, 640))
    t_pre = timeit(lambda: preprocess_batch(m, frames, (640, 640)))
    t_fwd = timeit(lambda: m.run_plan(plan))
    z = m.run_plan(plan)
    t_nms = timeit(lambda: nms_face_device(z, 0.4, 0.5))
    out, cnt, _, over = nms_face_device(z, 0.4, 0.5)
    cand = float(((z[..., 4] > 0.4) & (z[..., 4] * z[..., 15] > 0.4)).sum(1).float().mean())
    alg = sum(plan.algorithmic_bytes(i) for i in range(plan.n_ops))
    rate = B / (t_pre + t_fwd + t_nms)
    rec = {"config": f"{name}-face 640x640 batch {B} + batched NMS", "letterbox_ms": round(t_pre * 1e3, 3),
           "forward_decode_ms": round(t_fwd * 1e3, 3), "nms_ms": round(t_nms * 1e3, 3),
           "img_per_s": round(rate, 1), "frac_of_roofline": round(rate / ROOF_YOLO_IMG_S[name], 4),
           "cand_per_img": round(cand, 1), "dets_per_img": round(float(cnt.float().mean()), 2),
           "overflow": int(over.sum()), "algorithmic_GBps_forward": round(alg / t_fwd / 1e9, 1), "n_ops": plan.n_ops}
    print(json.dumps(rec), file=sys.stderr, flush=True)
    return rec


def yolo_to_embed(dev, B=256, faces_per_frame=4.0, cands=(2, 4, 6, 8, 12, 16), two_streams=True):
    """BASELINE configs[3]: YOLOv5s-face detect -> fmt = 1 crops -> Mobile-FaceNet 112x112, ~1024 crops per step
    (the candidate count is searched off the clock so that ~faces_per_frame boxes per frame survive NMS and the area
    filter)."""
    from face_detection_and_recognition_amd.pipeline import FacePipeline
    frames = W.make_frames(B, dev, seed=5)
    calib = frames                      # calibrate the candidate count on the measured batch itself (off the clock)
    emb = W.build_embedder(dev)
    best, table = None, []
    for cand in cands:
        det = W.build_yolo_detector(dev, calib, "yolov5s", cand_per_frame=cand)
        pipe = FacePipeline(det, emb, None, max_faces_per_frame=64)
        n = pipe.step(frames)["n_faces"] / float(B)
        table.append((cand, round(n, 2)))
        if best is None or abs(n - faces_per_frame) < abs(best[0] - faces_per_frame):
            best = (n, cand, pipe)
    _, cand, pipe = best
    n = pipe.step(frames)["n_faces"]
    t = timeit(lambda: pipe.step(frames))
    # frames at 19.2 k img/s + crops at 310 k crops/s = the step's time at the roofline
    t_roof = B / ROOF_YOLO_IMG_S["yolov5s"] + n / ROOF_MFN_CROPS_S
    rec = {"config": f"yolov5s-face detect -> Mobile-FaceNet 112x112, batch {B} frames", "ms": round(t * 1e3, 3),
           "crops_per_step": n, "cand_per_frame": cand, "search": table, "frames_per_s": round(B / t, 1),
           "crops_per_s": round(n / t, 1), "frac_of_roofline": round(t_roof / t, 4)}
    if not two_streams:
        print(json.dumps(rec), file=sys.stderr, flush=True)
        return rec
    # the same steps software-pipelined on two streams (FacePipeline.step_overlapped: the detector of batch k + 1 beside the
    # embedder of batch k), as bench.py runs configs[1]
    det2 = W.build_yolo_detector(dev, calib, "yolov5s", cand_per_frame=cand)
    pipe2 = FacePipeline(det2, emb, None, max_faces_per_frame=64, two_streams=True)

    def overlapped(k):
        for _ in range(k):
            pipe2.step_overlapped(frames)
        return pipe2.flush()["n_faces"]
    n2 = overlapped(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    overlapped(10)
    torch.cuda.synchronize()
    t2 = (time.perf_counter() - t0) / 10
    rec.update({"two_stream_ms": round(t2 * 1e3, 3), "two_stream_frames_per_s": round(B / t2, 1),
                "two_stream_crops_per_step": n2, "two_stream_frac_of_roofline": round(t_roof / t2, 4)})
    print(json.dumps(rec), file=sys.stderr, flush=True)
    return rec


def embed_1024(dev):
    emb = W.build_embedder(dev)
    plan = emb.plan_for(1024)
    plan.input.normal_()
    t = timeit(lambda: plan.run())
    alg = sum(plan.algorithmic_bytes(i) for i in range(plan.n_ops))
    rec = {"config": "Mobile-FaceNet 112x112 batch 1024 crops", "ms": round(t * 1e3, 3), "crops_per_s": round(1024 / t, 1),
           "frac_of_roofline": round(1024 / t / ROOF_MFN_CROPS_S, 4), "algorithmic_GBps": round(alg / t / 1e9, 1)}
    print(json.dumps(rec), file=sys.stderr, flush=True)
    return rec


def cosine(dev, M, Nr=10000, D=512):
    g = torch.Generator(device=dev).manual_seed(42)
    G = torch.randn((M, D), device=dev, generator=g)
    R = torch.randn((Nr, D), device=dev, generator=g)
    ginv, rinv = S.row_inv_norm(G), S.row_inv_norm(R)
    r3 = S.split3_rows(R)      # the reference set is split once (csrc/split.h), like its inverse norms
    t = timeit(lambda: S.cosine_filter(G, R, 0.3, ginv, rinv, r3=r3), n=3, warm=1)
    tf = 2.0 * M * Nr * D / t / 1e12
    rec = {"config": f"cosine filter {M} x {Nr} x {D}", "ms": round(t * 1e3, 2), "TFLOPs": round(tf, 1),
           "frac_of_roofline": round(tf / ROOF_FP32_MFMA_TF, 4), "pair_scores_per_s": round(M * Nr / t, 0)}
    print(json.dumps(rec), file=sys.stderr, flush=True)
    return rec


def jpeg_front_end(dev, n=128, threads=None):
    """JPEG decode of n synthetic 576 x 1024 4:2:0 frames -- host Huffman threads + device reconstruction
    (modules/utils/jpeg.py) -- against Pillow's (libjpeg-turbo) full decode on the same host threads."""
    import io
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from face_detection_and_recognition_amd.modules.utils import jpeg as J
    threads = threads or max(1, min(16, len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16"))))
    rng = np.random.default_rng(0)
    datas = []
    for _ in range(4):
        img = np.clip(np.cumsum(np.cumsum(rng.normal(0, 2.5, (576, 1024, 3)), 0), 1) * 0.2 + rng.normal(128, 20, (576, 1024, 3)),
                      0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, subsampling=2)
        datas.append(b.getvalue())
    datas = [datas[i % 4] for i in range(n)]

    def pil_all():
        with ThreadPoolExecutor(threads) as pool:
            return list(pool.map(lambda d: np.asarray(Image.open(io.BytesIO(d)).convert("RGB")), datas))
    t_gpu = timeit(lambda: J.decode_jpeg_batch(datas, dev, threads=threads), n=2, warm=1)
    t0 = time.perf_counter()
    pil_all()
    t_pil = time.perf_counter() - t0
    info, coefs = J.entropy_decode(datas[0], pinned=True)
    cd = coefs.to(dev)
    out = J.reconstruct(info, cd, dev)
    t_dev = timeit(lambda: J.reconstruct(info, cd, dev, out=out), n=50, warm=5)
    rec = {"config": f"JPEG decode, {n} frames 576x1024 4:2:0 q90 ({sum(len(d) for d in datas) // n // 1024} KiB each), "
                     f"{threads} host threads", "frames_per_s": round(n / t_gpu, 1), "ms": round(t_gpu * 1e3, 2),
           "pillow_frames_per_s": round(n / t_pil, 1), "device_half_us_per_frame": round(t_dev * 1e6, 1),
           "frac_of_roofline": None,
           "note": "host-bound: Huffman decoding on the host threads, everything after it on the GPU; byte-identical to libjpeg-turbo"}
    print(json.dumps(rec), file=sys.stderr, flush=True)
    return rec


def other_configs(dev):
    """The short legs bench.py appends to its result line (`other_configs`): BASELINE configs[2], configs[3], the embedder
    at batch 1024 and the configs[4] per-GPU shard, each with its fraction of SURVEY 8(d)'s roofline."""
    out = []
    for fn in (lambda: yolo("yolov5n", dev), lambda: yolo_to_embed(dev, cands=(4, 6), two_streams=False),
               lambda: embed_1024(dev), lambda: cosine(dev, 125_000), lambda: jpeg_front_end(dev)):
        out.append(fn())
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    which = sys.argv[1:] or ["yolov5n", "yolov5s", "c4", "embed", "cosine"]
    recs = []
    if "yolov5n" in which:
        recs.append(yolo("yolov5n", dev))
    if "yolov5s" in which:
        recs.append(yolo("yolov5s", dev))
    if "c4" in which:
        recs.append(yolo_to_embed(dev))
    if "embed" in which:
        recs.append(embed_1024(dev))
    if "cosine" in which:
        recs.append(cosine(dev, 125_000))
        recs.append(cosine(dev, 1_000_000))
    if "jpeg" in which or len(sys.argv) == 1:
        recs.append(jpeg_front_end(dev, 256))
    for r in recs:
        print(json.dumps(r), flush=True)

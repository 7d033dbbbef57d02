#!/usr/bin/env python3
"""Headline benchmark: faces/sec end-to-end (detect -> embed -> cosine-filter) on synthetic 576x1024 frames,
batch 256 per GPU (BASELINE.json configs[1]: BlazeFace back-camera 256^2 -> Mobile-FaceNet 112^2 -> cosine filter).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of 256 frames already resident in HBM: letterbox+normalise ->
BlazeFace-back forward -> anchor decode -> weighted NMS -> detections-to-crops -> crop/resize/normalise ->
Mobile-FaceNet -> cosine filter against a 10k-row reference set (N > 1: + RCCL all_gather of the step's embedding
matrix).  Frames shard by image: every rank processes its own 256 frames (weak scaling), no other collective.
Rank 0 prints ONE JSON line (contract in the task statement) including `roofline` (dominant kernel, HIP events
recorded on the launch stream during the timed steps) and `cpu_baseline` (the oracle timed on the host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from face_detection_and_recognition_amd import workload as W  # noqa: E402
from face_detection_and_recognition_amd.pipeline import FacePipeline  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
B_FRAMES = 256
N_REF = 10000
EMB_CAP_ROWS = 4096        # all_gather buffer rows per rank (>= faces per step per rank)
FRAME_BYTES = 576 * 1024 * 3


def host_cores():
    """Threads for the CPU baseline: the box's CPU share (16 per GPU on the pool), never more than the affinity."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline(det_model, emb_model, ref, frames_cpu, tau, budget_s=25.0):
    """The oracle (CPU restatement of the reference, torch-CPU fp32 + numpy) on a bounded sample of the same
    workload, all host cores.  Baseline only: it is never the thing shipped or the target."""
    import numpy as np
    from oracle import blazeface_ref, image_ref, mobilefacenet_ref, similarity_ref
    torch.set_num_threads(host_cores())
    sd_det = {k: v.detach().cpu() for k, v in det_model.net.state_dict().items()}
    sd_emb = {k: v.detach().cpu() for k, v in emb_model.state_dict().items()}
    anchors = det_model.net.anchors.cpu()
    refn = ref.cpu().numpy()
    frames = frames_cpu.numpy()
    iw, ih = det_model.input_size
    t0 = time.perf_counter()
    n_faces = n_frames = 0
    embs = []
    with torch.no_grad():
        for f in frames:                                            # the reference runs one frame per call
            if time.perf_counter() - t0 > budget_s:
                break
            n_frames += 1
            lb = image_ref.pad_resize_image(f, (iw, ih))[..., ::-1].copy()
            x = torch.from_numpy(lb).permute(2, 0, 1).unsqueeze(0)
            faces, _ = blazeface_ref.predict_on_batch(sd_det, x, anchors, True)
            d = faces[0].numpy()
            if len(d) == 0:
                continue
            d = d[:, [1, 0, 3, 2] + list(range(4, 17))]
            post = image_ref.dets_to_boxes(d.copy(), (f.shape[1], f.shape[0]), (iw, ih), det_model.det_thres,
                                           det_model.bbox_area_thres)
            for box in post["boxes"]:
                crop, _ = image_ref.crop_face(f, box)
                if crop.size == 0:
                    continue
                face = image_ref.mfn_lut()[image_ref.resize_bilinear_u8(crop, (112, 112))]
                xin = torch.from_numpy(np.ascontiguousarray(face.transpose(2, 0, 1))).unsqueeze(0)
                embs.append(mobilefacenet_ref.forward(sd_emb, xin)[0].numpy())
                n_faces += 1
        if embs:
            similarity_ref.cosine_filter(np.stack(embs), refn, tau)
    dt = time.perf_counter() - t0
    return n_faces / dt, n_faces, n_frames, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu-frames", type=int, default=96, help="frames in the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hooks (single-GPU rehearsal of the multi-rank path): BENCH_DIST_BACKEND=gloo BENCH_SAME_DEVICE=1
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    # ---- workload (off the clock) ----
    frames = W.make_frames(B_FRAMES, dev, seed=1234 + rank)
    det = W.build_detector(dev, W.make_frames(64, dev, seed=999))   # same calibration on every rank
    emb = W.build_embedder(dev)
    ref = W.make_reference(N_REF, dev)
    pipe = FacePipeline(det, emb, ref, tau=0.3)
    gather_buf = gather_out = None
    if world > 1:
        gather_buf = torch.zeros((EMB_CAP_ROWS, emb.embedding_size), device=dev)
        gather_out = torch.empty((world * EMB_CAP_ROWS, emb.embedding_size), device=dev)

    def step():
        out = pipe.step(frames)
        if world > 1:
            n = out["n_faces"]
            gather_buf[:n].copy_(out["emb"])
            dist.all_gather_into_tensor(gather_out, gather_buf)     # RCCL over xGMI: the step's embedding matrix
        return out["n_faces"]

    for _ in range(max(args.warmup, 1)):
        nf = step()
    torch.cuda.synchronize()

    # ---- per-op timers for the roofline figures ----
    # HIP events around an op cost a little, so: one un-timed probe step with events on every op finds the
    # dominant kernel family; the timed steps then carry events only on that family's launches.
    det_plan = det.net.plan_for(B_FRAMES)
    n_pad = (nf + pipe.bucket - 1) // pipe.bucket * pipe.bucket
    emb_plan = emb.plan_for(n_pad)
    plans = {"blazeface": det_plan, "mobilefacenet": emb_plan}
    probe = {k: p.new_timer() for k, p in plans.items()}
    for name, p in plans.items():
        p._timing = (probe[name], bytes([1] * p.n_ops))
    step()
    torch.cuda.synchronize()
    fam_ms = {}
    for name, p in plans.items():
        p._timing = None
        ms0 = (ctypes.c_float * p.n_ops)()
        p.accumulate(probe[name], ms0)
        p.destroy_timer(probe[name])
        for i in range(p.n_ops):
            fam_ms[p.kernel_name(i)] = fam_ms.get(p.kernel_name(i), 0.0) + ms0[i]
    dom = max(fam_ms, key=fam_ms.get)
    probe_share = fam_ms[dom] / sum(fam_ms.values())
    timers = {k: [p.new_timer() for _ in range(args.steps)] for k, p in plans.items()}
    masks = {k: bytes([1 if p.kernel_name(i) == dom else 0 for i in range(p.n_ops)]) for k, p in plans.items()}

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    faces = 0
    for k in range(args.steps):
        for name, p in plans.items():
            p._timing = (timers[name][k], masks[name])
        faces += step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    for p in plans.values():
        p._timing = None

    elapsed = t1 - t0
    tot = torch.tensor([elapsed, float(faces)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tot[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        fsum = tot[1:].clone()
        dist.all_reduce(fsum, op=dist.ReduceOp.SUM)
        elapsed, faces_all = float(tmax), float(fsum)
    else:
        faces_all = float(faces)

    # ---- roofline of the dominant kernel (rank 0) ----
    roof = None
    if rank == 0:
        ms_tot, launches, bytes_tot = 0.0, 0, 0
        for name, p in plans.items():
            ms = (ctypes.c_float * p.n_ops)()
            for t in timers[name]:
                p.accumulate(t, ms)
                p.destroy_timer(t)
            for i in range(p.n_ops):
                if masks[name][i]:
                    ms_tot += ms[i]
                    launches += args.steps
                    bytes_tot += p.algorithmic_bytes(i) * args.steps
        achieved = bytes_tot / (ms_tot * 1e-3) / 1e9 if ms_tot > 0 else 0.0
        # HBM bytes per launch from the committed PMC passes of this same command (profiles/*_pmc_traffic.json,
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 FETCH x2 correction), if present
        traffic = None
        try:
            import glob
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]
            traffic = json.load(open(latest)).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        # whole-pipeline view: op-granular bytes of both networks + letterbox per step over the step time
        pipe_bytes = sum(p.algorithmic_bytes(i) for p in plans.values() for i in range(p.n_ops)) + \
            B_FRAMES * (FRAME_BYTES + 256 * 256 * 3 * 4)
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "pipeline_algorithmic_GBps": round(pipe_bytes / (elapsed / args.steps) / 1e9, 1),
                "pipeline_frac": round(pipe_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                "avg_launch_us": round(ms_tot * 1e3 / launches, 2), "launches_per_step": launches // args.steps,
                "algorithmic_bytes_per_launch": int(bytes_tot // launches),
                "share_of_network_kernel_time": round(probe_share, 3)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        v, n_cpu, nfr, dt = cpu_baseline(det, emb, ref, frames[:args.cpu_frames].cpu(), pipe.tau)
        cpu = {"value": round(v, 2), "unit": "faces/s", "cores": host_cores(), "kind": "port",
               "sample": f"first {nfr} frames of the same batch (time-boxed), one frame per call like the reference: "
                         f"{n_cpu} faces in {dt:.1f} s, torch-CPU fp32 oracle"}

    if rank == 0:
        line = {
            "metric": "faces/sec end-to-end (detect->embed->cosine-filter), 576x1024 batch=256",
            "value": round(faces_all / elapsed, 1), "unit": "faces/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BlazeFace back-camera 256x256, batch 256 synthetic 576x1024 frames per GPU -> "
                                   "weighted NMS -> Mobile-FaceNet 112x112 -> cosine filter vs 10k x 512 reference",
                       "frames_per_step_per_gpu": B_FRAMES, "faces_per_frame": round(faces_all / world / args.steps / B_FRAMES, 3),
                       "frames_per_s": round(B_FRAMES * world * args.steps / elapsed, 1), "n_ref": N_REF,
                       "weights": "seeded synthetic (no weights ship with the reference)",
                       "parallelism": f"frames sharded by image, {world} rank(s), 1 per GPU" +
                                      (", all_gather of the step's embeddings over RCCL" if world > 1 else "")},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Batched detect -> crop -> embed -> similarity-filter pipeline, resident on one GPU.

This is the reference's own composition (face_extraction/extract_faces_from_dataset.py:270-307:
``net.inf_func`` -> ``bbox_conf_area_func`` -> crop with offsets -> ``get_face_features``) followed by the
similarity filter, run for a whole batch of frames at once: every stage is a HIP kernel launched on the
caller's stream and the only host round trip is reading the number of faces found (it sizes the embedder batch).
"""
import numpy as np
import os

import torch

from . import _lib as L
from . import similarity as S
from .modules.mobile_facenet.utils import crops_to_input, mfn_lut
from .modules.utils.image import letterbox_geometry

FACE_OFFSETS = (-6, -1, 4, 5)   # tx, ty, bx, by  (extract_faces_from_dataset.py:285-287)


def scale_coords_params(in_size, orig_size):
    """gain / pad of scale_coords (modules/utils/image.py:83-87) as fp32 (numpy promotes python floats to the
    float32 array dtype)."""
    iw, ih = in_size
    w, h = orig_size
    gain = min(ih / h, iw / w)
    pad_x, pad_y = (iw - w * gain) / 2, (ih - h * gain) / 2
    return np.float32(gain), np.float32(pad_x), np.float32(pad_y)


class FacePipeline:
    """detector: a BlazeFaceModel or YOLOV5FaceModel (HIP); embedder: a HIP MobileFaceNet;
    reference: (Nr, E) CUDA tensor of reference embeddings for the cosine filter (or None)."""

    # embed(): a batch a little above a multiple of ROUND_CROPS crops is run as that multiple + the remainder on a side stream
    ROUND_CROPS = 512     # crops whose tiles fill whole rounds of workgroups in every Depth_Wise kernel (2 / 4 / 7 tiles per crop, 512 slots)
    TAIL_MAX = 96         # largest remainder worth splitting off (measured: tools/lab/embed_split_probe.py)
    TAIL_CAP = 128        # capacity of the remainder's plan

    def __init__(self, detector, embedder, reference=None, tau=0.3, max_faces_per_frame=8, bucket=8, two_streams=False,
                 split_tail=True):
        self.det = detector
        self.emb = embedder
        self.tau = float(tau)
        self.max_faces_per_frame = int(max_faces_per_frame)
        self.bucket = int(bucket)
        self.dev = embedder._device()
        self.lut = mfn_lut(self.dev)
        # step_overlapped with two_streams: embed + filter of batch k run on a SIDE stream beside the detector of batch
        # k + 1 (the split-MFMA embedder kernels are matrix-core bound, the BlazeFace kernels vector-ALU / HBM bound, and
        # every kernel's last, partly empty round of workgroups is filled by the other stream's work)
        prio = int(os.environ.get("FP_EMB_STREAM_PRIO", "0"))     # lab knob (tools/lab/README.md): -1 = high priority
        self.emb_stream = torch.cuda.Stream(device=self.dev, priority=prio) if two_streams else None
        self.split_tail = bool(split_tail)
        self.tail_stream = torch.cuda.Stream(device=self.dev) if split_tail else None
        net = getattr(detector, "net", None)
        self._co_net = net if hasattr(net, "co_scheduled") else None     # BlazeFace: no whole-CU ops beside the embedder's kernels
        self.set_reference(reference)

    def set_reference(self, reference):
        self.reference = None if reference is None else reference.to(self.dev, torch.float32).contiguous()
        self.rinv = None if reference is None else S.row_inv_norm(self.reference)
        # the reference set split once into bf16 planes for the split-MFMA cosine kernel (similarity.split3_rows)
        self.ref3 = None if reference is None or self.reference.shape[1] % 32 else S.split3_rows(self.reference)

    # -- stages ----------------------------------------------------------------------------------
    def detect(self, frames, max_det=-1, beside=False):
        """frames (B, H, W, 3) u8 BGR on device -> (dets, counts, overflow or None).  max_det: -1 = the detector's
        default cap, None = uncapped (the exact re-run after an overflow).  beside: the detector's kernels will run
        beside the embedder's on the other stream (step_overlapped with two_streams)."""
        if self._co_net is not None:      # selects the plan (blazeface.py plan_for): several pipelines may share one detector
            self._co_net.co_scheduled = bool(beside)
        out = self.det.raw_batch(frames) if max_det == -1 else self.det.raw_batch(frames, max_det=max_det)
        return out if len(out) == 3 else (out[0], out[1], None)

    def crops(self, frames, dets, counts):
        """Device-side B7 + crop arithmetic -> (items, info, n_faces tensor)."""
        lib = L.load()
        B, H, W, _ = frames.shape
        cap = B * self.max_faces_per_frame
        items = torch.empty((cap, 9), dtype=torch.int32, device=self.dev)
        info = torch.empty((cap, 7), dtype=torch.float32, device=self.dev)
        nf = torch.empty((1,), dtype=torch.int32, device=self.dev)
        iw, ih = self.det.input_size
        gain, px, py = scale_coords_params((iw, ih), (W, H))
        fmt = getattr(self.det, "dets_fmt", 0)
        row = dets.shape[-1]
        tx, ty, bx, by = FACE_OFFSETS
        L.check(lib.fp_dets_to_crops(L.ptr(dets), L.ptr(counts), B, dets.shape[1], row, fmt, iw, ih, W, H,
                                     float(self.det.det_thres), float(self.det.bbox_area_thres), float(gain),
                                     float(px), float(py), tx, ty, bx, by, 112, 112, cap, L.ptr(items), L.ptr(info),
                                     L.ptr(nf), L.current_stream(self.dev)), "fp_dets_to_crops")
        return items, info, nf

    def embed(self, frames, items, n_faces):
        """Crop + resize + normalise into the embedder's input, run Mobile-FaceNet.  -> (n_faces, E), a view into the
        embedder plan's arena.  ONE plan (arena sized for the largest batch seen, in steps of 256 crops) serves every
        face count: it runs on the first n_pad = n_faces rounded up to `bucket` images (8 keeps the 14x14 layers'
        row count a multiple of the 32-row MFMA tiles the streaming 1x1 kernels need), so at most bucket - 1 crops of
        work are padding and a varying face count neither builds new plans nor pins new arenas."""
        if n_faces == 0:
            return torch.zeros((0, self.emb.embedding_size), device=self.dev)
        n_pad = (n_faces + self.bucket - 1) // self.bucket * self.bucket
        cap = max(getattr(self, "_emb_cap", 0), (n_pad + 255) // 256 * 256)
        self._emb_cap = cap
        plan = self.emb.plan_for(cap, n_run=n_pad)
        self.emb_key, self.emb_n_pad = (cap, n_pad), n_pad     # the plan itself stays owned by the embedder's LRU cache
        main = n_pad // self.ROUND_CROPS * self.ROUND_CROPS
        if self.split_tail and main and 0 < n_pad - main <= self.TAIL_MAX and main < n_faces:
            return self._embed_split(frames, items, n_faces, n_pad, main, plan)
        crops_to_input(frames, items, n_faces, plan.input, self.lut)
        if n_pad > n_faces:
            plan.input[n_faces:n_pad].zero_()    # padding crops: defined inputs (every op is per-image, their rows are dropped)
        plan.run(n=n_pad)
        return plan.out[:n_faces]

    def _embed_split(self, frames, items, n_faces, n_pad, main, plan):
        """~528 crops are 1056 band tiles of a 14 x 14 Depth_Wise kernel on 512 workgroup slots: two full rounds and a third
        with 32 tiles, in EVERY launch (28 x 28: 2112 tiles, 56 x 56: 3696) -- the last 16 crops cost 0.25 ms of a 2.3 ms
        forward.  The first `main` crops (whole rounds in every kernel) run on the current stream, the remainder as its own
        small forward on a side stream beside them (its own plan / arena): 2.30 -> 2.16 ms alone on the GPU.  Every op is
        per-image and kernels do not depend on the batch, so the rows are bit-identical to the one-run form
        (tests: test_embedder_split_tail_is_bit_identical).  Returns a new (n_faces, E) tensor."""
        rem, rem_pad = n_faces - main, n_pad - main
        tail = self.emb.plan_for(self.TAIL_CAP, n_run=rem_pad)
        cur = torch.cuda.current_stream(self.dev)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self.tail_stream):
            self.tail_stream.wait_event(ready)           # items / frames are complete (and the tail plan's previous consumer is done)
            frames.record_stream(self.tail_stream)
            items.record_stream(self.tail_stream)
            crops_to_input(frames, items[main:], rem, tail.input, self.lut)
            if rem_pad > rem:
                tail.input[rem:rem_pad].zero_()
            tail.run(n=rem_pad)
            done = torch.cuda.Event()
            done.record(self.tail_stream)
        crops_to_input(frames, items, main, plan.input, self.lut)
        plan.run(n=main)
        cur.wait_event(done)
        return torch.cat([plan.out[:main], tail.out[:rem]])

    @property
    def emb_plan(self):
        """The embedder plan of the last step (looked up in the embedder's plan cache; not held by the pipeline)."""
        cap, n_pad = self.emb_key
        return self.emb.plan_for(cap, n_run=n_pad)

    # -- software-pipelined form ----------------------------------------------------------------------
    def step_overlapped(self, frames):
        """step() with the host round trip taken off the GPU's critical path: this call ENQUEUES the detector stages of
        `frames` and then finishes the PREVIOUS call's batch (embed + filter), whose face count -- produced a whole
        detector pass ago -- is already on the host when it is read (pinned buffer + event).  The GPU queue never drains
        while the host waits.  Returns the previous batch's result dict (None on the first call); flush() returns the
        last one.  Same kernels, same numbers as step(); a detector overflow (more survivors than the cap in some frame)
        falls back to the exact un-capped re-run for that batch."""
        dets, counts, over = self.detect(frames, beside=self.emb_stream is not None)
        items, info, nf = self.crops(frames, dets, counts)
        # two pinned count buffers, used alternately: at most one batch is pending while the previous one's is read
        if getattr(self, "_host_counts", None) is None:
            self._host_counts, self._host_k = [torch.empty((2,), dtype=torch.int32).pin_memory() for _ in range(2)], 0
        host = self._host_counts[self._host_k & 1]
        self._host_k += 1
        both = nf if over is None else torch.stack([nf[0], over.sum().to(torch.int32)])
        host[:both.numel()].copy_(both, non_blocking=True)
        if over is None:
            host[1] = 0
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        prev, self._pending = getattr(self, "_pending", None), (frames, items, info, host, ev)
        return None if prev is None else self._finish(prev)

    def flush(self):
        """Finish the batch a previous step_overlapped() call left pending (None if there is none)."""
        prev, self._pending = getattr(self, "_pending", None), None
        return None if prev is None else self._finish(prev)

    def _finish(self, pending):
        frames, items, info, host, ev = pending
        ev.synchronize()
        n, n_over = int(host[0]), int(host[1])
        if n_over:                              # > MAX_DET survivors in some frame: exact re-run without a cap
            dets, counts, _ = self.detect(frames, max_det=None)
            items, info, nf = self.crops(frames, dets, counts)
            n = int(nf.item())
        cap = items.shape[0]
        if n > cap:
            raise L.FacepathError(f"{n} faces in the batch exceed max_faces_per_frame*B = {cap}")
        if self.emb_stream is None:
            emb = self.embed(frames, items, n)
            res = self.filter(emb)
            out = dict(n_faces=n, info=info[:n], emb=emb.clone(), items=items[:n])
        else:
            main = torch.cuda.current_stream(self.dev)
            self.emb_stream.wait_event(ev)                 # the crops of this batch (detector stream)
            if getattr(self, "_emb_done", None) is not None:
                main.wait_event(self._emb_done)            # (results of the batch before are complete for the caller)
            with torch.cuda.stream(self.emb_stream):
                for t in (frames, items, info):
                    t.record_stream(self.emb_stream)
                emb = self.embed(frames, items, n)
                res = self.filter(emb)
                out = dict(n_faces=n, info=info[:n], emb=emb.clone(), items=items[:n])
                # allocated on the side stream, consumed by the caller on the main stream (after `done`): tell the caching
                # allocator, or it hands the blocks to the next embed / filter while main-stream reads are still queued
                for t in (out["emb"],) + (tuple(res) if res is not None else ()):
                    t.record_stream(main)
                self._emb_done = torch.cuda.Event()
                self._emb_done.record(self.emb_stream)
            out["done"] = self._emb_done                   # the caller waits for this event before it reads the results
        if res is not None:
            out.update(best=res[0], arg=res[1], keep=res[2])
        return out

    def filter(self, emb):
        if self.reference is None or emb.shape[0] == 0:
            return None
        return S.cosine_filter(emb, self.reference, self.tau, rinv=self.rinv, r3=self.ref3)

    # -- whole step ------------------------------------------------------------------------------
    def step(self, frames, beside=False):
        """One pass over a batch of frames.  Returns dict(n_faces, info, emb, best, arg, keep).
        ``emb`` is a copy (the embedder's output lives in its plan arena and the next step overwrites it).
        beside: use the detector plan of the two-stream steps (measurement: bench.py's per-op probe pass)."""
        dets, counts, over = self.detect(frames, beside=beside)
        items, info, nf = self.crops(frames, dets, counts)
        # the one host sync of the step: the face count (sizes the embedder batch) and the detector's overflow flag
        if over is None:
            n, n_over = int(nf.item()), 0
        else:
            n, n_over = torch.stack([nf[0], over.sum().to(torch.int32)]).tolist()
        if n_over:                              # > MAX_DET survivors in some frame: exact re-run without a cap
            dets, counts, _ = self.detect(frames, max_det=None)
            items, info, nf = self.crops(frames, dets, counts)
            n = int(nf.item())
        cap = items.shape[0]
        if n > cap:
            raise L.FacepathError(f"{n} faces in the batch exceed max_faces_per_frame*B = {cap}")
        emb = self.embed(frames, items, n)
        res = self.filter(emb)
        out = dict(n_faces=n, info=info[:n], emb=emb.clone(), items=items[:n])
        if res is not None:
            out.update(best=res[0], arg=res[1], keep=res[2])
        return out

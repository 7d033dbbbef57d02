"""Lab: what would the bench step cost if a kernel family were FREE?  Runs bench.py's two-stream (or single-stream) steps with
the ops of the named families removed from the detector / embedder plans (results are wrong: timing only) -- an upper bound of
what any rewrite of that family can buy in the step the bench times.
    python tools/lab/ablate_ops.py [--overlap 2|0] fam1 fam2 ...        (family = substring of the kernel name; "none" first)"""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from face_detection_and_recognition_amd import _lib as L, workload as W
from face_detection_and_recognition_amd.pipeline import FacePipeline

def main():
    args = sys.argv[1:]
    overlap = 2
    if args and args[0] == "--overlap":
        overlap = int(args[1]); args = args[2:]
    fams = ["none"] + args
    dev = torch.device("cuda:0")
    batches = [W.make_frames(256, dev, seed=1234 + b) for b in range(4)]
    det = W.build_detector(dev, W.make_frames(64, dev, seed=999))
    emb = W.build_embedder(dev)
    ref = W.make_reference(10000, dev)
    pipe = FacePipeline(det, emb, ref, tau=0.3, two_streams=overlap == 2)

    state = {"k": 0, "cache": None}
    net = det.net
    orig_post = net.postprocess

    def post(r, c):
        """the ablated detector plan leaves garbage in r / c: post-process the raw outputs a correct run left for this batch
        (same decode + NMS work, same faces for the embedder)"""
        k = state["k"] % 4
        state["k"] += 1
        if state["cache"] is None:
            return orig_post(r, c)
        if len(state["cache"]) < 4:
            state["cache"].append((r.clone(), c.clone()))
            return orig_post(r, c)
        return orig_post(*state["cache"][k])
    net.postprocess = post

    def run(count):
        state["k"] = 0
        ns = []
        if overlap:
            for i in range(count):
                out = pipe.step_overlapped(batches[i % 4])
                if out is not None: ns.append(out["n_faces"])
            ns.append(pipe.flush()["n_faces"])
        else:
            for i in range(count): ns.append(pipe.step(batches[i % 4])["n_faces"])
        return ns
    run(8); torch.cuda.synchronize()
    state["cache"] = []
    run(4); torch.cuda.synchronize()       # (fills the cache: batches 0..3 in order)
    assert len(state["cache"]) == 4
    plans = [det.net.last_plan, pipe.emb_plan]
    saved = [(p.ops, p.n_ops) for p in plans]
    for fam in fams:
        for p, (ops, n) in zip(plans, saved):
            keep = [ops[i] for i in range(n) if fam == "none" or fam not in p.lib.fp_op_kernel_name(L.C.byref(ops[i])).decode()]
            p.ops = (L.FpOp * max(len(keep), 1))(*keep); p.n_ops = len(keep)
            if p is not plans[0]:
                p.n_run = -1                  # force set_batch to rewrite the N of the swapped-in ops
        run(4); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); run(40); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 40)
        removed = sum(n - p.n_ops for p, (ops, n) in zip(plans, saved))
        print(f"without {fam:32s} ({removed:2d} ops removed): {best * 1e3:.3f} ms per step", flush=True)
    for p, (ops, n) in zip(plans, saved):
        p.ops, p.n_ops = ops, n
        if p is not plans[0]:
            p.n_run = -1

main()

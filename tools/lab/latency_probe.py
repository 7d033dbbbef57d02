import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from face_detection_and_recognition_amd import workload as W
dev = torch.device("cuda:0")
frames = W.make_frames(8, dev)
det = W.build_detector(dev, frames)
emb = W.build_embedder(dev)
for name, plan in (("blazeface B=1", det.net.plan_for(1)), ("mobilefacenet N=1", emb.plan_for(1)), ("mobilefacenet N=64", emb.plan_for(64))):
    for _ in range(5): plan.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        plan.run(); torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / 200 * 1e3
    t0 = time.perf_counter()
    for _ in range(200): plan.run()
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / 200 * 1e3
    print(f"{name}: {plan.n_ops} ops, latency {lat:.3f} ms (sync each), back-to-back {thr:.3f} ms")

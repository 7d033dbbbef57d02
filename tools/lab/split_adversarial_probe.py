"""Probe: error of pwx6_kernel / cosine_x6_kernel on adversarial operands vs the fp32 fmaf chain and torch fp32 (fp64 truth).
   python tools/lab/split_adversarial_probe.py"""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_parity import _adversarial_f32, _fmaf_chain
from face_detection_and_recognition_amd.plan import CompiledPlan, PlanBuilder
dev = torch.device("cuda:0")
for kind in ("ones", "ties", "positive", "normal"):
    for k in (128, 512, 1152):
        rng = np.random.default_rng(k + len(kind))
        N, H, W, n = 2, 16, 16, 128
        if kind == "normal":
            x = rng.normal(0, 1, (N, k, H, W)).astype(np.float32); w = rng.normal(0, 1, (n, k, 1, 1)).astype(np.float32)
        else:
            x = _adversarial_f32(rng, (N, k, H, W), kind); w = _adversarial_f32(rng, (n, k, 1, 1), kind)
        pb = PlanBuilder(N); xb, ob = pb.new_buf(H, W, k), pb.new_buf(H, W, n); pb.conv(xb.view(), w, ob.view())
        plan = CompiledPlan(pb, dev)
        plan.buf_tensor(xb, N).copy_(torch.from_numpy(x).to(dev).permute(0, 2, 3, 1)); plan.run(); torch.cuda.synchronize()
        got = plan.buf_tensor(ob, N).cpu().numpy().reshape(-1, n)
        a = np.ascontiguousarray(x.transpose(0, 2, 3, 1)).reshape(-1, k); b = np.ascontiguousarray(w.reshape(n, k).T)
        exact = a.astype(np.float64) @ b.astype(np.float64)
        den = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
        t32 = F.conv2d(torch.from_numpy(x), torch.from_numpy(w)).permute(0, 2, 3, 1).reshape(-1, n).numpy()
        ch = _fmaf_chain(a, b)
        u = 2.0 ** -24
        f = lambda v: (np.abs(v - exact) / den)
        print(f"{kind:9s} K={k:5d} {plan.kernel_name(0):28s} max/u: x6 {f(got).max()/u:7.2f} chain {f(ch).max()/u:7.2f} torch {f(t32).max()/u:7.2f} | rms/u: x6 {np.sqrt((f(got)**2).mean())/u:6.3f} chain {np.sqrt((f(ch)**2).mean())/u:6.3f} torch {np.sqrt((f(t32)**2).mean())/u:6.3f} | bias/u x6 {((got-exact)/den).mean()/u:7.3f} chain {((ch-exact)/den).mean()/u:7.3f}", flush=True)

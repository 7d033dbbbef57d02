"""Lab: the embedder's last, nearly empty round of workgroups.  Mobile-FaceNet on n crops as ONE run against a run on the first
512 * floor(n / 512) crops + a concurrent run of the remainder on a second stream (own plan / arena).
    python tools/lab/embed_split_probe.py [n ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from face_detection_and_recognition_amd import workload as W

dev = torch.device("cuda:0")
emb = W.build_embedder(dev)
side = torch.cuda.Stream(device=dev)

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e3

for n in [int(a) for a in sys.argv[1:]] or [528, 520, 536, 560, 600, 768, 1024, 1040]:
    cap = (n + 255) // 256 * 256
    big = emb.plan_for(cap, n_run=n)
    big.input.normal_()
    t_one = timeit(lambda: big.run(n=n))
    main_n = n // 512 * 512
    rem = n - main_n
    if main_n == 0 or rem == 0:
        print(f"n={n}: one run {t_one:.3f} ms (no split)"); continue
    small = emb.plan_for(512 if cap != 512 else 256, n_run=rem)
    assert small is not big
    small.input.normal_()
    def split():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            small.run(n=rem)
            done = torch.cuda.Event(); done.record(side)
        big.run(n=main_n)
        torch.cuda.current_stream().wait_event(done)
    t_split = timeit(split)
    t_main = timeit(lambda: big.run(n=main_n))
    t_rem = timeit(lambda: small.run(n=rem))
    print(f"n={n}: one run {t_one:.3f} ms | split {main_n}+{rem} on two streams {t_split:.3f} ms | main alone {t_main:.3f}, remainder alone {t_rem:.3f}", flush=True)

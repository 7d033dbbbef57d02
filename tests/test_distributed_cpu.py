"""world_size-2 gloo tests (CPU) of the multi-GPU data path: frame sharding and the one exchange step."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from face_detection_and_recognition_amd.distributed import shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 256, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from face_detection_and_recognition_amd import distributed as D
    from oracle import similarity_ref
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    G = rng.normal(0, 1, (101, 64)).astype(np.float32)
    R = rng.normal(0, 1, (37, 64)).astype(np.float32)
    g0, g1 = D.shard_range(len(G), rank, world)
    r0, r1 = D.shard_range(len(R), rank, world)

    def filter_fn(g, r, tau):      # the checker stands in for the HIP kernel on CPU tensors
        b, a, k, _ = similarity_ref.cosine_filter(g.numpy(), r.numpy(), tau)
        return torch.from_numpy(b), torch.from_numpy(a), torch.from_numpy(k)

    best, arg, keep = D.sharded_cosine_filter(torch.from_numpy(G[g0:g1]), torch.from_numpy(R[r0:r1]), 0.1, filter_fn)
    rb, ra, rk, _ = similarity_ref.cosine_filter(G, R, 0.1)
    ok = np.allclose(best.numpy(), rb[g0:g1], atol=1e-6) and np.array_equal(arg.numpy(), ra[g0:g1]) and \
        np.array_equal(keep.numpy(), rk[g0:g1])
    rows, offs = D.all_gather_rows(torch.from_numpy(R[r0:r1]))
    ok = ok and np.array_equal(rows.numpy(), R) and offs[-1] == len(R)
    mean = D.sharded_l2_mean(torch.from_numpy(R[r0:r1]))
    ok = ok and np.allclose(mean.numpy(), R.mean(0), atol=1e-6)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_sharded_similarity_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]

"""Multi-GPU data path: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

Frames are independent through letterbox -> detect -> NMS -> crop -> embed, so they shard by image with no
collective.  The similarity stage is the only exchange:
  * cosine filter of a row-sharded gallery against a reference set that was *produced* sharded:
    one all_gather of the (small) reference block, then each rank filters its own gallery rows;
  * l2_mean mode: all_reduce(sum) of the per-rank partial sums / counts of the reference rows, then local filtering.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block of items owned by ``rank`` (blocks differ by at most one item)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_rows(local_rows, group=None):
    """Gather row blocks of different lengths from every rank -> (all_rows, offsets).
    Two collectives: the row counts (one int64 per rank) and the padded blocks."""
    world = dist.get_world_size(group)
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    padded[:local_rows.shape[0]] = local_rows
    out = torch.empty((world * cap,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    rows = torch.cat([out[r * cap: r * cap + counts[r]] for r in range(world)])
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + c)
    return rows, offsets


def sharded_cosine_filter(local_gallery, local_reference, tau, filter_fn, group=None):
    """Each rank holds a gallery shard and the reference rows it embedded; returns this rank's
    (best, arg, keep) against the FULL reference set (arg indexes the gathered reference)."""
    full_ref, _ = all_gather_rows(local_reference, group)
    return filter_fn(local_gallery, full_ref, tau)


def sharded_l2_mean(local_reference, group=None):
    """Class mean over reference rows spread across ranks: all_reduce of (sum, count)."""
    s = local_reference.sum(dim=0)
    n = torch.tensor([float(local_reference.shape[0])], dtype=s.dtype, device=s.device)
    dist.all_reduce(s, group=group)
    dist.all_reduce(n, group=group)
    return s / n

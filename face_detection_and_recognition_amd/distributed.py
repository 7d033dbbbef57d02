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
    Two collectives: the row counts (one int64 per rank, read on the host in ONE transfer -- the result is ragged,
    so the host has to know the sizes) and the padded blocks."""
    world = dist.get_world_size(group)
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
    counts_t = torch.empty((world,), dtype=torch.int64, device=local_rows.device)
    dist.all_gather_into_tensor(counts_t, n, group=group)
    counts = counts_t.tolist()
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


def all_gather_blocks(block, n_valid, group=None):
    """Variable-length gather with NO host round trip: every rank contributes a fixed-capacity block (cap, D) of which
    the first ``n_valid`` rows (a 1-element device tensor) are meaningful.  Returns
      rows  (world * cap, D)  -- rank r's rows at [r * cap, r * cap + count_r)
      valid (world * cap,)    -- bool mask of the meaningful rows, built on device from the gathered counts
      counts (world,) int64 device tensor.
    Callers multiply their per-row quantities (inverse norms) by ``valid`` instead of compacting."""
    world = dist.get_world_size(group)
    cap = block.shape[0]
    counts = torch.empty((world,), dtype=torch.int64, device=block.device)
    dist.all_gather_into_tensor(counts, n_valid.to(torch.int64).reshape(1), group=group)
    rows = torch.empty((world * cap,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(rows, block.contiguous(), group=group)
    idx = torch.arange(world * cap, device=block.device)
    valid = (idx % cap) < counts.repeat_interleave(cap)
    return rows, valid, counts


def cross_rank_match(block, n_valid, tau, filter_fn, inv_norm_fn, group=None, n_rows=None):
    """The pairwise-similarity exchange of the detect -> embed path (north_star: all-gather of the final embedding
    matrix): every rank's faces of this step against the faces found by all OTHER ranks in the same step.
    block (cap, D): this rank's embeddings, first n_valid rows meaningful (n_valid: 1-element device tensor; a count
    above cap means the rank had more faces than the block holds -- every row of the block is meaningful then and the
    caller re-runs with a larger block, StepExchange).
    filter_fn(G, R, tau, rinv) -> (best, arg, keep) is the cosine filter (HIP: similarity.cosine_filter);
    inv_norm_fn(R) -> (rows,) inverse row norms.  Own rows and padding rows take part with inverse norm 0
    (score exactly 0), so nothing is compacted and nothing is read on the host.  n_rows: match only the first n_rows
    local rows (the host knows its own face count; the default matches all cap rows).  Returns (best, arg, keep) for
    those local rows (entries past n_valid are padding) and the gathered counts; arg indexes the gathered matrix
    (rank * cap + i).  A row whose row maximum landed on a masked column -- every cosine against the other ranks' faces
    is negative, or no other rank found a face -- gets arg = -1, keep = False and best = -1 (a lower bound: the true
    maximum is negative and the kernel does not return it)."""
    rank = dist.get_rank(group)
    cap = block.shape[0]
    rows, valid, counts = all_gather_blocks(block, n_valid, group)
    others = valid.clone()
    others[rank * cap:(rank + 1) * cap] = False
    rinv = inv_norm_fn(rows) * others.to(rows.dtype)
    local = block if n_rows is None else block[:n_rows]
    best, arg, keep = (t.to(others.device) for t in filter_fn(local, rows, tau, rinv))   # (a rehearsal's kernel may run elsewhere)
    hit = others[arg.long().clamp_(0, others.shape[0] - 1)]          # did the maximum land on a real peer row?
    arg = torch.where(hit, arg, torch.full_like(arg, -1))
    best = torch.where(hit, best, torch.full_like(best, -1.0))
    keep = keep & hit
    return best, arg, keep, counts


class StepExchange:
    """cross_rank_match of step k running BESIDE step k + 1's detector: the exchange is issued on a side stream (the
    collectives of torch's NCCL / RCCL backend follow the stream that is current when they are called) behind an event
    on the step's embeddings, and its result is handed out one step late.  Without it every rank waits for the slowest
    rank's embeddings inside every step; with it a rank only waits when it is a whole step ahead.  Two buffer sets, so
    step k + 2 reuses the block of step k only after that exchange has finished (event wait on the main stream).

    The gather blocks have a fixed capacity (all_gather needs equal blocks and no rank knows the others' face counts
    in advance).  The capacity GROWS: every exchange gathers the true counts, and when a hand-out finds a count above the
    capacity that exchange used, every rank -- they all read the same gathered counts, so they agree without another
    message -- enlarges its blocks to fit (rounded up to `grow` rows), re-runs that step's exchange in place and keeps the
    larger capacity.  Nothing is dropped and nothing raises; `cap` is only the starting size.

    On a CPU process group (gloo tests) there are no streams: the exchange runs in place, the hand-out order is the
    same."""

    def __init__(self, cap, dim, device, tau, filter_fn, inv_norm_fn, group=None, comm=lambda t: t, grow=256, pad_rows=8):
        self.tau, self.filter_fn, self.inv_norm_fn, self.group, self.comm = tau, filter_fn, inv_norm_fn, group, comm
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.cap, self.dim, self.grow, self.pad_rows = int(cap), int(dim), int(grow), int(pad_rows)
        self.regrown = 0                                      # exchanges that had to be re-run with a larger block
        self.blocks = [torch.zeros((self.cap, dim), device=self.device) for _ in range(2)]
        self.counts = [torch.zeros((1,), dtype=torch.int64, device=self.device) for _ in range(2)]
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.done = [torch.cuda.Event() if self.cuda else None for _ in range(2)]
        self.results = [None, None]
        self.embs = [None, None]                              # the step's embeddings, kept for a re-run at a larger capacity
        self.host_counts = [None, None]                       # the gathered counts on the host (pinned), filled behind `done`
        self.k = 0

    def _match(self, slot):
        """Queue slot's exchange on the current stream: block copy (first min(n, cap) rows), gather, match, counts to host."""
        emb, n = self.embs[slot]
        if self.blocks[slot].shape[0] != self.cap:
            self.blocks[slot] = torch.zeros((self.cap, self.dim), device=self.device)
        m = min(n, self.cap)
        self.blocks[slot][:m].copy_(emb[:m])
        self.counts[slot].fill_(n)
        n_rows = min(self.cap, (m + self.pad_rows - 1) // self.pad_rows * self.pad_rows)
        res = cross_rank_match(self.comm(self.blocks[slot]), self.comm(self.counts[slot]), self.tau,
                               self.filter_fn, self.inv_norm_fn, self.group, n_rows=max(n_rows, self.pad_rows))
        counts = res[3]
        if counts.is_cuda:
            if self.host_counts[slot] is None:
                self.host_counts[slot] = torch.empty(counts.shape, dtype=counts.dtype).pin_memory()
            self.host_counts[slot].copy_(counts, non_blocking=True)
        else:
            self.host_counts[slot] = counts
        return res + (self.cap,)

    def submit(self, emb, n):
        """Queue the exchange of this step's embeddings (emb: (n, dim) on `device`; the caller must not overwrite it).
        Returns the result of the PREVIOUS step's exchange -- (best, arg, keep, counts, cap), ready for use on the current
        stream; arg indexes rank * cap + row -- or None on the first step."""
        slot = self.k & 1
        self.k += 1
        self.embs[slot] = (emb, int(n))
        if self.cuda:
            main = torch.cuda.current_stream(self.device)
            if self.results[slot] is not None:
                main.wait_event(self.done[slot])           # the exchange that last used this slot (two steps ago)
            ready = torch.cuda.Event()
            ready.record(main)
            emb.record_stream(self.stream)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                self.results[slot] = self._match(slot)
                self.done[slot].record(self.stream)
        else:
            self.results[slot] = self._match(slot)
        return self._take(slot ^ 1) if self.k > 1 else None

    def _take(self, slot):
        res = self.results[slot]
        if res is None:
            return None
        if self.cuda:
            self.done[slot].synchronize()                  # (an exchange queued a whole step ago)
        need = int(self.host_counts[slot].max())
        if need > res[4]:
            # some rank found more faces than the block held.  Every rank sees the same counts: all grow, all re-run.
            self.cap = max(self.cap, (need + self.grow - 1) // self.grow * self.grow)
            self.regrown += 1
            res = self.results[slot] = self._match(slot)   # on the current stream, complete before the caller's next op
            if self.cuda:
                self.done[slot].record(torch.cuda.current_stream(self.device))
        elif self.cuda:
            main = torch.cuda.current_stream(self.device)
            main.wait_event(self.done[slot])
            for t in res:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(main)                  # produced on the side stream, consumed here
        return res

    def drain(self):
        """The result of the last submitted step (call once after the final submit; makes the current stream wait)."""
        return self._take((self.k - 1) & 1) if self.k else None


def sharded_cosine_filter(local_gallery, local_reference, tau, filter_fn, group=None, equal_blocks=False):
    """Each rank holds a gallery shard and the reference rows it embedded; returns this rank's
    (best, arg, keep) against the FULL reference set (arg indexes the gathered reference).
    equal_blocks: every rank's reference block has the same number of rows (BASELINE configs[4]: 10 k rows / world)
    -> ONE all_gather_into_tensor and no size exchange."""
    if equal_blocks:
        world = dist.get_world_size(group)
        full_ref = torch.empty((world * local_reference.shape[0],) + tuple(local_reference.shape[1:]),
                               dtype=local_reference.dtype, device=local_reference.device)
        dist.all_gather_into_tensor(full_ref, local_reference.contiguous(), group=group)
    else:
        full_ref, _ = all_gather_rows(local_reference, group)
    return filter_fn(local_gallery, full_ref, tau)


def sharded_l2_mean(local_reference, group=None):
    """Class mean over reference rows spread across ranks: all_reduce of (sum, count)."""
    s = local_reference.sum(dim=0)
    n = torch.tensor([float(local_reference.shape[0])], dtype=s.dtype, device=s.device)
    dist.all_reduce(s, group=group)
    dist.all_reduce(n, group=group)
    return s / n

"""Frame-parallel sharding across the GPUs of a node: no data-path collective.

Frames are independent units (inference-mode BatchNorm, static anchors), so the
hot path shards by frame: rank r of W owns a contiguous block of frame ids and
its own engine handle.  The only cross-rank traffic is the barrier around the
timed region and a MAX-reduce of the elapsed time (bench.py contract); backend
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""


def rank_frames(rank, world, frames_per_rank):
    """Global frame ids owned by `rank` (contiguous block, weak scaling)."""
    if not (0 <= rank < world) or frames_per_rank < 0:
        raise ValueError("bad rank / world / frames_per_rank")
    start = rank * frames_per_rank
    return list(range(start, start + frames_per_rank))


def split_frames(n_frames, world):
    """Strong-scaling split of n_frames over `world` ranks: contiguous, sizes differ by <= 1."""
    base, extra = divmod(n_frames, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of a scalar (elapsed seconds).  dist is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(count, dist=None, device=None):
    """All ranks' frame counts (for whole-job throughput = sum(frames) / max(time))."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(count)]
    import torch
    t = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=device if device is not None else "cpu")
    t[dist.get_rank()] = int(count)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]

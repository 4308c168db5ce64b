"""Multi-GPU sharding for the batch entry points: one process per GPU (torch.distributed; backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The path shards by independent blobs: contiguous blob ranges per rank,
setup tables replicated per GPU, NO collective on the data path.  The only communication is the
gather of the small per-blob outputs (48 B per commitment; 268,288 B per blob of cells+proofs)
and, for batch verification, an AND/MAX reduction of the per-shard verdict and return code.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world):
    """Contiguous, balanced [lo, hi) ranges: the first n % world ranks get one extra unit."""
    base, extra = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def gather_rows(local, n_total, group=None):
    """All-gather row-sharded results (uint8 tensor [n_local, width]) into [n_total, width] on every
    rank.  Shards may be ragged by one row, so they are padded to the largest shard."""
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_total, world)
    width = local.shape[1]
    max_rows = max(hi - lo for lo, hi in bounds)
    padded = torch.zeros((max_rows, width), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([parts[r][: hi - lo] for r, (lo, hi) in enumerate(bounds)], dim=0)


_GATHER_UNSUPPORTED = False


def gather_to_rank0(local, group=None):
    """The "trivial gather" of the sharded commitment path: every rank's [rows, width] results (equal shards) to rank 0,
    which gets [world * rows, width]; the other ranks get None.  One RCCL gather over xGMI ("nccl" backend, device
    tensors, enqueued behind the producing work: the caller goes on); gloo (CPU tests) gathers host copies.
    A backend build without `gather` raises the same error on every rank: from then on the rows travel by all-gather
    (every rank receives them, rank 0 keeps them) -- the same bytes arrive at rank 0 either way."""
    global _GATHER_UNSUPPORTED
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if dist.get_backend(group) != "nccl":
        local = local.cpu()
    if not _GATHER_UNSUPPORTED:
        try:
            parts = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
            dist.gather(local, parts, dst=0, group=group)
            return torch.cat(parts, dim=0) if rank == 0 else None
        except (RuntimeError, NotImplementedError) as e:
            if "gather" not in str(e).lower() and "support" not in str(e).lower():
                raise
            _GATHER_UNSUPPORTED = True
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)
    return torch.cat(parts, dim=0) if rank == 0 else None


def sharded_map(compute, n_total, width, device, group=None):
    """Run compute(lo, hi) -> uint8 tensor [hi-lo, width] on this rank's shard and gather."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(n_total, world)[rank]
    local = compute(lo, hi) if hi > lo else torch.empty((0, width), dtype=torch.uint8, device=device)
    return gather_rows(local.to(device), n_total, group)


def sharded_verify(verify, n_total, device, group=None):
    """verify(lo, hi) -> (ret, ok) on each shard; the batch is valid iff every shard is.
    Returns (max ret over shards, AND of ok) on every rank.  A shard's random-linear-combination
    challenge differs from the single-call one, the verdict does not (soundness error 2^-255)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(n_total, world)[rank]
    ret, ok = verify(lo, hi) if hi > lo else (0, True)
    t = torch.tensor([ret, 0 if ok else 1], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t[0].item()), int(t[1].item()) == 0

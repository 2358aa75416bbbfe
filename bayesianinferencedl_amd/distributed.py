"""Multi-GPU plumbing: samples are independent, so the path shards with NO data-path
collective; operators are replicated by each rank building its own handles;
the only exchange is one gather of the per-sample results at the end.  One process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests)."""
from __future__ import annotations


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous block of ceil(total / world) samples per rank: [lo, hi)."""
    per = -(-total // world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def gather_rows(local, world: int, total: int | None = None, force: bool = False):
    """All ranks contribute ``local`` [rows_r, d] (torch tensor; every rank but the last must hold
    ceil(total/world) rows); returns the concatenation [total, d] on every rank.  ``force``: run the
    collective even in a one-rank group (the RCCL rehearsal of bench.py --force-pg)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
        return local
    per = local.shape[0]
    if total is not None:
        per = -(-total // world)
        if local.shape[0] < per:                         # last shard may be short: pad, trim after
            pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], 0)
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out if total is None else out[:total]

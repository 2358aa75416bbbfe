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


def gather_rows(local, world: int, total: int | None = None, force: bool = False, async_op: bool = False):
    """All ranks contribute ``local`` [rows_r, d] (torch tensor; every rank but the last must hold
    ceil(total/world) rows); returns the concatenation [total, d] on every rank.  ``force``: run the
    collective even in a one-rank group (the RCCL rehearsal of bench.py --force-pg).
    ``async_op``: returns (result, work) -- the collective runs on the backend's own stream behind what is queued on the
    current one, and the caller's later kernels do not wait for it; ``work.wait()`` (None where no collective ran) makes the
    current stream wait.  The caller keeps ``local`` and the result alive until then."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
        return (local, None) if async_op else local
    per = local.shape[0]
    if total is not None:
        per = -(-total // world)
        if local.shape[0] < per:                         # last shard may be short: pad, trim after
            pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            local = torch.cat([local, pad], 0)
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if async_op:
        local = local.contiguous()
        work = dist.all_gather_into_tensor(out, local, async_op=True)
        return (out if total is None else out[:total]), work
    dist.all_gather_into_tensor(out, local.contiguous())
    return out if total is None else out[:total]


class FinromComm:
    """The same gather through the C ABI (finrom_comm_* in include/finrom.h): RCCL loaded by the library itself, no
    torch.distributed.  rank 0: ``uid = FinromComm.unique_id()`` and hand the bytes to the other ranks; every rank (its device
    current): ``comm = FinromComm(rank, world, uid)``; ``comm.gather(ptr, count, out_ptr)`` all-gathers `count` doubles per rank
    between DEVICE buffers (DeviceBuffer.ptr / tensor.data_ptr())."""

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from ._ffi import check, lib
        buf = C.create_string_buffer(128)
        check(lib().finrom_comm_unique_id(buf), "finrom_comm_unique_id")
        return buf.raw

    def __init__(self, rank: int, world: int, uid: bytes):
        import ctypes as C
        from ._ffi import check, lib
        if len(uid) != 128:
            raise ValueError("unique id must be FINROM_COMM_ID_BYTES = 128 bytes")
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        check(lib().finrom_comm_init(C.byref(self._h), rank, world, C.create_string_buffer(uid, 128)), "finrom_comm_init")

    def gather(self, send_ptr, count, recv_ptr, stream=None):
        from ._ffi import check, lib
        check(lib().finrom_gather(self._h, send_ptr, int(count), recv_ptr, stream), "finrom_gather")

    def close(self):
        from ._ffi import lib
        if getattr(self, "_h", None):
            lib().finrom_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

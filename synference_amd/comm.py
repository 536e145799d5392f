"""RCCL communicator of the data-parallel training path (include/synference_hip.h, ``sf_comm_*``; csrc/sf_comm.hip).

One process per GPU.  The process group of ``torch.distributed`` is the CONTROL plane here -- rendezvous, the
broadcast of the seed / parameters / resume decision, the two scalars per epoch -- and it also ships RCCL's unique id to the
ranks; the per-step gradient all-reduce itself runs inside the library's epoch call (``sf_flow_train_epoch_dp``) on the
library's stream, through a communicator created here.  The reference has no counterpart (it trains on CPU threads:
examples/sbi/slurm/train_final_model.slurm:26).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib

ID_BYTES = 128   # SF_COMM_ID_BYTES
_bound = False


def _bind_library() -> None:
    """Name the RCCL to bind before the first sf_comm_* call: SF_RCCL_LIB if set, else the librccl.so that PyTorch-ROCm ships
    (so that this process holds ONE RCCL, the one torch.distributed's "nccl" backend uses), else the library's own search."""
    global _bound
    if _bound:
        return
    _bound = True
    if os.environ.get("SF_RCCL_LIB"):
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(cand):
        _lib.load().sf_comm_set_library(cand.encode())   # (an error here only means RCCL is bound already)


def library_info() -> dict:
    """{"path": ..., "version": ...} of the RCCL the library bound (binds it if necessary)."""
    _bind_library()
    buf, ver = C.create_string_buffer(1024), C.c_int(0)
    _lib.check(_lib.load().sf_comm_library(buf, len(buf), C.byref(ver)))
    return {"path": buf.value.decode(), "version": int(ver.value)}


class RcclComm:
    """An RCCL communicator over the ranks of the default process group (or of this process alone)."""

    def __init__(self, handle: int, nranks: int, rank: int, device: torch.device):
        self.handle, self.nranks, self.rank, self.device = handle, nranks, rank, device

    @classmethod
    def create(cls, device, nranks: int = 1, rank: int = 0, exchange=None) -> "RcclComm":
        """``exchange(id_bytes or None) -> id_bytes``: rank 0 passes the id in and every rank gets it back (default: one
        ``broadcast_object_list`` over the default process group; not called for a single rank)."""
        device = torch.device(device)
        torch.cuda.set_device(device)
        uid = C.create_string_buffer(ID_BYTES)
        # Everything that can fail on ONE rank (binding the library, drawing the id) is done before the first collective, and its
        # outcome travels WITH the id: a rank that raised here while its peers waited in the broadcast -- or in ncclCommInitRank --
        # would leave them there for good.  Every rank learns of a failure anywhere and raises the same error.
        err = None
        lib = None
        try:
            _bind_library()
            lib = _lib.load()
            library_info()
            if rank == 0:
                _lib.check(lib.sf_comm_unique_id(uid, ID_BYTES))
        except Exception as e:   # noqa: BLE001 -- reported below, on every rank
            err = f"rank {rank}: {type(e).__name__}: {e}"
        raw = bytes(uid.raw)
        if nranks > 1:
            if exchange is None:
                import torch.distributed as dist

                def exchange(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]

                def agree(msg):
                    msgs = [None] * dist.get_world_size()
                    dist.all_gather_object(msgs, msg)
                    return next((m for m in msgs if m), None)
            else:
                def agree(msg):
                    return msg
            raw = exchange((raw, err) if rank == 0 else None)
            raw, err0 = raw if isinstance(raw, tuple) else (raw, None)
            err = agree(err or err0)
        if err:
            raise RuntimeError(f"RCCL communicator not created ({err})")
        h = C.c_void_p()
        _lib.check(lib.sf_comm_create(raw, ID_BYTES, int(nranks), int(rank), C.byref(h)))
        return cls(h.value, int(nranks), int(rank), device)

    @classmethod
    def from_process_group(cls, device) -> "RcclComm":
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return cls.create(device, dist.get_world_size(), dist.get_rank())
        return cls.create(device, 1, 0)

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks of a contiguous float32 device tensor, on torch's current stream."""
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            raise ValueError("RcclComm.all_reduce_ needs a contiguous float32 tensor on the communicator's device")
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().sf_comm_all_reduce_sum(C.c_void_p(self.handle), C.c_void_p(t.data_ptr()), t.numel(), st))
        return t

    def close(self) -> None:
        if self.handle:
            _lib.load().sf_comm_destroy(C.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_cached: Optional[RcclComm] = None


def default_comm(device) -> Optional[RcclComm]:
    """The communicator ``train_flow`` uses under a process group whose backend is RCCL ("nccl"): created once per process and
    group size, None when there is no such group (gloo rehearsals keep the host-staged per-step path).  ``SF_DP_FUSED=0``
    disables it (A-B runs against the per-step c10d all-reduce)."""
    global _cached
    import torch.distributed as dist
    if os.environ.get("SF_DP_FUSED", "1") == "0":
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl" or dist.get_world_size() < 2:
        return None
    device = torch.device(device)
    if device.type != "cuda":
        return None
    if _cached is None or _cached.handle is None or _cached.nranks != dist.get_world_size() or _cached.device != device:
        _cached = RcclComm.from_process_group(device)
    return _cached

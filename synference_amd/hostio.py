"""Device -> host hand-over of posterior draws in the reference's container.

``SBI_Fitter.sample_posterior`` returns a host float64 array (N, S, D) (ref: src/synference/sbi_runner.py:6436: the
reference allocates float64 and fills it galaxy by galaxy from ``.cpu().numpy()`` copies, 6442-6457).  The draws are fp32 on
the device; moving them costs more than drawing them (2e6 draws x 5 parameters: 1.9 ms of kernel, 40 MB over PCIe, 80 MB of
float64 written on the host), so the hand-over is the library's native pipeline ``sf_copy_to_host_f64``
(csrc/sf_hostio.hip: pinned staging ring on a copy stream, widening by a pool of host threads with streaming stores, copy and
widening overlapped); this module owns the RESULT buffers and the call.  PCIe carries fp32 (half the bytes of a device-side
``.double()``); nothing touches the values.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Optional

import numpy as np
import torch

from . import _lib

_lock = threading.Lock()
_bg: Optional[ThreadPoolExecutor] = None


def usable_cores() -> int:
    """CPU share of this process: cgroup quota when there is one (a GPU box shows 256 cores and grants 16), else the
    affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# ---- result buffers -------------------------------------------------------------------------------------------------
# A fresh 80 MB numpy array costs ~6 ms of first-touch page faults (mmap'd by malloc, zeroed by the kernel page by page)
# -- four times the kernel that draws the samples -- and glibc hands such blocks back to the OS on free (its mmap threshold
# tops out at 32 MB), so every call would pay again.  The arrays handed to callers are therefore remembered here, and one
# of them is handed out again ONLY when nothing outside this list refers to it any more (the caller dropped the previous
# result: reference count of the list entry alone).  A result a caller still holds -- or any view of it -- is never touched.
_results: "list[np.ndarray]" = []
_RESULTS_MAX = 3
_RESULTS_MAX_BYTES = 2 << 30


def result_array(shape) -> np.ndarray:
    """Uninitialised float64 C-contiguous array of ``shape``: a recycled result buffer when one is free, else new."""
    import sys
    shape = tuple(int(v) for v in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * 8
    if os.environ.get("SF_HOSTIO_POOL", "1") == "0" or nbytes < (8 << 20):
        return np.empty(shape, dtype=np.float64)
    with _lock:
        for i in range(len(_results)):
            # references: the list slot + getrefcount's own argument = 2 when nobody else holds the array or a view of it
            if _results[i].nbytes == nbytes and sys.getrefcount(_results[i]) == 2:
                arr = _results.pop(i)
                arr.shape = shape
                _results.append(arr)
                return arr
        arr = np.empty(shape, dtype=np.float64)
        if nbytes <= _RESULTS_MAX_BYTES:
            _results.append(arr)
            while len(_results) > _RESULTS_MAX or sum(a.nbytes for a in _results) > _RESULTS_MAX_BYTES:
                _results.pop(0)
        return arr


# ---- pinned result buffers: the sampler writes the reference's float64 host container itself (sf_flow_set_sample_output_f64) ----
# hipHostMalloc of 80 MB costs milliseconds, so the pinned buffers are recycled by the same rule as the plain ones: a buffer
# is handed out again only when neither the array nor a view of it is referenced outside this list.
_pinned: "list[tuple[torch.Tensor, np.ndarray]]" = []
_PINNED_MAX_BYTES = 2 << 30      # larger results (configs[4]: 4 GB) take the staged copy instead of pinning that much host memory


def pinned_result(shape):
    """(pinned float64 CPU tensor, numpy view of it) of ``shape``, or None when the result is too large / too small to be worth
    pinning or the pool is switched off (``SF_HOSTIO_PINNED=0``).  The numpy array keeps the tensor's storage alive."""
    import sys
    shape = tuple(int(v) for v in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * 8
    if os.environ.get("SF_HOSTIO_PINNED", "1") == "0" or nbytes < (1 << 20) or nbytes > _PINNED_MAX_BYTES or not torch.cuda.is_available():
        return None
    with _lock:
        for i in range(len(_pinned)):
            t, arr = _pinned[i]
            # references to the array: the tuple in the list + getrefcount's argument (+ this loop's `arr`) = 3 when free
            if t.numel() * 8 == nbytes and sys.getrefcount(arr) == 3:
                _pinned.pop(i)
                t = t.view(shape)
                arr = t.numpy()
                _pinned.append((t, arr))
                return t, arr
        try:
            t = torch.empty(shape, dtype=torch.float64, pin_memory=True)
        except RuntimeError:
            return None
        arr = t.numpy()
        _pinned.append((t, arr))
        while len(_pinned) > _RESULTS_MAX or sum(p[0].numel() * 8 for p in _pinned) > 2 * _PINNED_MAX_BYTES:
            _pinned.pop(0)
        return t, arr


class PendingCopy:
    """Hand-over in flight (``to_host_f64(..., wait=False)``): ``result()`` blocks until the host array is complete."""

    def __init__(self, out, future, dev):
        self.out, self._future, self._dev = out, future, dev

    def result(self) -> np.ndarray:
        if self._future is not None:
            self._future.result()
            self._future = self._dev = None
        return self.out


def to_host_f64(dev: torch.Tensor, out: Optional[np.ndarray] = None, wait: bool = True):
    """float64 host copy of a float32 tensor (any shape), bit-identical to ``dev.double().cpu().numpy()``.  ``out``: optional
    float64 C-contiguous array of the same shape to fill.  ``wait=False`` runs the (blocking, GIL-free) native call on a helper
    thread and returns a ``PendingCopy``: the caller may launch the next chunk's kernels meanwhile."""
    global _bg
    if dev.dtype != torch.float32:
        raise ValueError("to_host_f64 takes a float32 tensor")
    shape = tuple(dev.shape)
    if out is None:
        out = result_array(shape)
    elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float64 array of the tensor's shape")
    if dev.numel() == 0:
        return out if wait else PendingCopy(out, None, None)
    if not dev.is_cuda:
        np.copyto(out, dev.detach().numpy(), casting="same_kind")
        return out if wait else PendingCopy(out, None, None)
    dev = dev.contiguous()
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(dev.device).cuda_stream)
    n = dev.numel()

    def run():
        with torch.cuda.device(dev.device):
            _lib.check(lib.sf_copy_to_host_f64(C.c_void_p(dev.data_ptr()), C.c_void_p(out.ctypes.data), n, stream))

    if wait:
        run()
        return out
    with _lock:
        if _bg is None:
            _bg = ThreadPoolExecutor(max_workers=1, thread_name_prefix="sf-hostio")
    return PendingCopy(out, _bg.submit(run), dev)

"""Device -> host hand-over of posterior draws in the reference's container.

``SBI_Fitter.sample_posterior`` returns a host float64 array (N, S, D) (ref: src/synference/sbi_runner.py:6436: the
reference allocates float64 and fills it galaxy by galaxy from ``.cpu().numpy()`` copies, 6442-6457).  The draws are fp32 on
the device; moving them costs more than drawing them (2e6 draws x 5 parameters: 1.8 ms of kernel, 40 MB over PCIe, 80 MB of
float64 written on the host), so the hand-over is a pipeline:

* the device tensor is cut into row chunks of a few MB; chunk k is copied D2H on a copy stream into one of a small ring of
  PINNED fp32 staging buffers (allocated once per process and reused: a fresh pinned allocation costs milliseconds);
* as soon as its copy event has completed, a pool thread widens the chunk fp32 -> float64 straight into the caller's result
  array (numpy releases the GIL in the cast loop; first-touch page faults of the fresh 80 MB array are spread over the pool
  as well), while chunk k + 1 is in flight on the bus.

PCIe carries fp32 (half the bytes of a device-side ``.double()``); nothing here touches the values.
"""
from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Optional

import numpy as np
import torch

_lock = threading.Lock()
_ring: "list[torch.Tensor]" = []
_ring_bytes = 0
_pool: Optional[ThreadPoolExecutor] = None
_pool_workers = 0
_copy_streams: "dict[int, torch.cuda.Stream]" = {}


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


def _get_pool(workers: Optional[int]) -> ThreadPoolExecutor:
    global _pool, _pool_workers
    w = int(workers) if workers else max(1, min(16, usable_cores()))
    with _lock:
        if _pool is None or _pool_workers != w:
            if _pool is not None:
                _pool.shutdown(wait=True)
            _pool = ThreadPoolExecutor(max_workers=w, thread_name_prefix="sf-hostio")
            _pool_workers = w
        return _pool


def _get_ring(n_buf: int, nbytes: int) -> "list[torch.Tensor]":
    global _ring, _ring_bytes
    with _lock:
        if len(_ring) < n_buf or _ring_bytes < nbytes:
            _ring = [torch.empty(nbytes // 4, dtype=torch.float32, pin_memory=True) for _ in range(n_buf)]
            _ring_bytes = nbytes
        return _ring[:n_buf]


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


def to_host_f64(dev: torch.Tensor, out: Optional[np.ndarray] = None, chunk_mb: float = 8.0, n_buf: int = 4,
                workers: Optional[int] = None) -> np.ndarray:
    """float64 host copy of a float32 tensor (any shape, first axis = rows), bit-identical to
    ``dev.double().cpu().numpy()``.  ``out``: optional float64 C-contiguous array of the same shape to fill."""
    if dev.dtype != torch.float32:
        raise ValueError("to_host_f64 takes a float32 tensor")
    shape = tuple(dev.shape)
    if out is None:
        out = result_array(shape)
    elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float64 array of the tensor's shape")
    if dev.numel() == 0:
        return out
    if not dev.is_cuda:
        np.copyto(out, dev.detach().numpy(), casting="same_kind")
        return out
    dev = dev.contiguous()
    n_rows = shape[0]
    row_elems = dev.numel() // n_rows
    flat_dev = dev.view(n_rows, row_elems)
    flat_out = out.reshape(n_rows, row_elems)
    rows_per = max(1, int(chunk_mb * (1 << 20)) // (4 * row_elems))
    rows_per = min(rows_per, n_rows)
    ring = _get_ring(n_buf, rows_per * row_elems * 4)
    pool = _get_pool(workers)
    di = dev.device.index if dev.device.index is not None else torch.cuda.current_device()
    cs = _copy_streams.get(di)
    if cs is None:
        cs = _copy_streams[di] = torch.cuda.Stream(device=dev.device)
    # the draws were written on the caller's current stream
    cs.wait_stream(torch.cuda.current_stream(dev.device))
    n_chunks = (n_rows + rows_per - 1) // rows_per
    # a chunk is widened by several threads (a 4 MB chunk on one thread would make the pool as slow as its slowest member)
    split = max(1, min(_pool_workers, 4))
    busy: "list[Optional[list]]" = [None] * n_buf   # futures still reading ring[b]

    def widen(dst: np.ndarray, src: np.ndarray, ev: torch.cuda.Event, first: bool):
        if first:
            ev.synchronize()
        np.copyto(dst, src, casting="same_kind")

    futures = []
    with torch.cuda.stream(cs):
        for k in range(n_chunks):
            b = k % n_buf
            if busy[b] is not None:          # the staging buffer is free once its widening is done
                for fu in busy[b]:
                    fu.result()
            r0, r1 = k * rows_per, min(n_rows, (k + 1) * rows_per)
            stage = ring[b][: (r1 - r0) * row_elems].view(r1 - r0, row_elems)
            stage.copy_(flat_dev[r0:r1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
            src = stage.numpy()
            fs = []
            n = r1 - r0
            for p in range(split):
                a0, a1 = (p * n) // split, ((p + 1) * n) // split
                if a1 > a0:
                    fs.append(pool.submit(widen, flat_out[r0 + a0: r0 + a1], src[a0:a1], ev, True))
            busy[b] = fs
            futures.extend(fs)
    for fu in futures:
        fu.result()
    # the device tensor must outlive the copies issued on the side stream
    dev.record_stream(cs)
    return out

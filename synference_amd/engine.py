"""``HipFlow``: one flow handle of the C ABI, fed with torch device tensors.

PyTorch is plumbing here (device memory, streams); every number is produced by the HIP
library.  There is no eager/CPU fallback: without the library or without a GPU the calls raise.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .spec import KIND_ID, FlowSpec, num_params


def retry_width(pending: int, attempt: int, max_attempts: int, total: int = 0) -> int:
    """Attempts evaluated per pending slot in a retry round (same rule as sf_flow_sample): speculate only
    up to the work a latency-bound round could do anyway (~2.6e5 items), at most 16 attempts per slot (the
    16-row MAF kernel resolves a slot within one 16-draw tile)."""
    A = 1
    if attempt > 0:
        budget = 262144
        while A < 16 and 2 * A * pending <= budget and attempt + 2 * A <= max_attempts:
            A *= 2
    return A


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32c(t, device) -> torch.Tensor:
    t = torch.as_tensor(t)
    return t.to(device=device, dtype=torch.float32).contiguous()


class HipFlow:
    """Owner of an ``sf_flow`` handle."""

    def __init__(self, spec: FlowSpec, device="cuda:0"):
        self.spec = spec
        self.lib = _lib.load()
        self.device = torch.device(device)
        self._keep = (spec.theta_mean, spec.theta_std, spec.x_mean, spec.x_std, spec.perms)
        d = _lib.sf_flow_desc(
            kind=KIND_ID[spec.kind], D=spec.D, C=spec.C, H=spec.H, T=spec.T, K=spec.K, NB=spec.NB,
            scale_fn=0 if spec.scale_fn == "softplus" else 1, hidden_bf16=1 if spec.hidden_bf16 else 0,
            tail_bound=spec.tail_bound, min_bin_width=spec.min_bin_width, min_bin_height=spec.min_bin_height,
            min_derivative=spec.min_derivative, maf_eps=spec.maf_eps, lu_eps=spec.lu_eps,
            theta_mean=spec.theta_mean.ctypes.data_as(_lib.c_f32p),
            theta_std=spec.theta_std.ctypes.data_as(_lib.c_f32p),
            x_mean=spec.x_mean.ctypes.data_as(_lib.c_f32p),
            x_std=spec.x_std.ctypes.data_as(_lib.c_f32p),
            perms=spec.perms.ctypes.data_as(_lib.c_i32p) if spec.kind == "maf" else None, ar_slope=spec.ar_slope)
        h = C.c_void_p()
        _lib.check(self.lib.sf_flow_create(C.byref(d), C.byref(h)))
        self.handle = h
        self.n_params = int(self.lib.sf_flow_num_params(h))
        assert self.n_params == num_params(spec), (self.n_params, num_params(spec))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            self.lib.sf_flow_destroy(h)
            self.handle = None

    # ---- host-only diagnostics (no GPU) ----------------------------------------------------
    def packed_size(self) -> int:
        return int(self.lib.sf_flow_packed_size(self.handle))

    def pack_table(self) -> Tuple[np.ndarray, np.ndarray]:
        n = self.packed_size()
        s1 = np.empty(n, np.int32)
        s2 = np.empty(n, np.int32)
        _lib.check(self.lib.sf_flow_pack_table(self.handle, s1.ctypes.data_as(_lib.c_i32p),
                                               s2.ctypes.data_as(_lib.c_i32p), n))
        return s1, s2

    def pack_table16(self) -> Tuple[np.ndarray, np.ndarray]:
        """Gather tables of the 16-row sampler image (empty arrays when the flow has none)."""
        n = int(self.lib.sf_flow_packed16_size(self.handle))
        s1 = np.empty(n, np.int32)
        s2 = np.empty(n, np.int32)
        if n:
            _lib.check(self.lib.sf_flow_pack_table16(self.handle, s1.ctypes.data_as(_lib.c_i32p),
                                                     s2.ctypes.data_as(_lib.c_i32p), n))
        return s1, s2

    def pack_table16b(self) -> np.ndarray:
        """Gather table of the split-bf16 hidden blocks of the persistent 16-row sampler (empty when the flow has none):
        entry i = logical index | (part << 30), part 0 = hi, 1 = lo; -1 = zero."""
        n = int(self.lib.sf_flow_packed16b_size(self.handle))
        s = np.empty(n, np.int32)
        if n:
            _lib.check(self.lib.sf_flow_pack_table16b(self.handle, s.ctypes.data_as(_lib.c_i32p), n))
        return s

    TRC_FIELDS = ("ok NT NI t_stride g_stride o_win o_b0 o_w1 o_b1 o_w2 o_b2 o_wf o_bf o_wfT o_w2T o_w1T o_winT "
                  "g_win g_b0 g_w1 g_b1 g_w2 g_b2 g_wf g_bf kend0 kend1 kend2 kend3 kbeg0 kbeg1 kbeg2 kbeg3 c_insrc c_jobs "
                  "n_jobs").split()

    NSC_FIELDS = ("ok NT NI OTQ KM kc_h kc_in0 kc_in1 kc_in2 t_stride g_stride o_win o_bin o_wg0 o_wg1 o_bg0 o_bg1 o_w10 o_w11 "
                  "o_b10 o_b11 o_w20 o_w21 o_b20 o_b21 o_wout o_bout o_lu o_woutT o_w2T0 o_w2T1 o_w1T0 o_w1T1 o_winT "
                  "g_win g_bin g_wg0 g_wg1 g_bg0 g_bg1 g_w10 g_w11 g_b10 g_b11 g_w20 g_w21 g_b20 g_b21 g_wout g_bout g_lu").split()

    def trainc_table(self):
        """Cooperative 16-row training image (csrc/sf_layout.h: SfTrcDev for a MAF, SfNscDev for an NSF): None when the
        flow has none, else (src1, src2, gdst, descriptor dict, constants image)."""
        n = int(self.lib.sf_flow_trainc_size(self.handle))
        if not n:
            return None
        s1 = np.empty(n, np.int32)
        s2 = np.empty(n, np.int32)
        gd = np.empty(self.n_params, np.int32)
        desc = np.zeros(64, np.int32)
        ncst = int(self.lib.sf_flow_cst_size(self.handle))
        cst = np.empty(ncst, np.float32)
        _lib.check(self.lib.sf_flow_trainc_table(self.handle, s1.ctypes.data, s2.ctypes.data, n, gd.ctypes.data, self.n_params,
                                                 desc.ctypes.data, cst.ctypes.data, ncst))
        d = {k: int(v) for k, v in zip(self.TRC_FIELDS if self.spec.kind == "maf" else self.NSC_FIELDS, desc)}
        d["n_grad"] = int(self.lib.sf_flow_trainc_grad_size(self.handle))
        return s1, s2, gd, d, cst

    def describe(self) -> dict:
        """Layout description of the handle (fixed at creation: parsed once)."""
        d = getattr(self, "_describe", None)
        if d is None:
            buf = C.create_string_buffer(1 << 16)
            _lib.check(self.lib.sf_flow_describe(self.handle, buf, len(buf)))
            d = self._describe = json.loads(buf.value.decode())
        return dict(d)

    # ---- device calls -----------------------------------------------------------------------
    def _dev(self):
        if not torch.cuda.is_available():
            raise RuntimeError("synference_amd: no GPU visible; the HIP flow engine has no CPU fallback")
        torch.cuda.set_device(self.device)

    def set_params(self, flat: torch.Tensor) -> None:
        self._dev()
        flat = _f32c(flat, self.device)
        _lib.check(self.lib.sf_flow_set_params(self.handle, _ptr(flat), flat.numel(), 1, _stream(self.device)))

    def get_params(self) -> torch.Tensor:
        """The logical parameter vector held by the handle (device tensor)."""
        self._dev()
        out = torch.empty(num_params(self.spec), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sf_flow_get_params(self.handle, _ptr(out), out.numel(), 1, _stream(self.device)))
        return out

    def log_prob(self, theta, x) -> torch.Tensor:
        self._dev()
        theta, x = _f32c(theta, self.device), _f32c(x, self.device)
        B = theta.shape[0]
        if theta.shape != (B, self.spec.D) or x.shape != (B, self.spec.C):
            raise ValueError(f"theta {tuple(theta.shape)} / x {tuple(x.shape)} do not match "
                             f"(B,{self.spec.D}) / (B,{self.spec.C})")
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sf_flow_log_prob(self.handle, _ptr(theta), _ptr(x), B, _ptr(out), _stream(self.device)))
        return out

    def inverse(self, z, x) -> Tuple[torch.Tensor, torch.Tensor]:
        self._dev()
        z, x = _f32c(z, self.device), _f32c(x, self.device)
        B = z.shape[0]
        if z.shape != (B, self.spec.D) or x.shape != (B, self.spec.C):
            raise ValueError("shape mismatch")
        th = torch.empty_like(z)
        ld = torch.empty(B, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sf_flow_inverse_from_noise(self.handle, _ptr(z), _ptr(x), B, _ptr(th), _ptr(ld),
                                                       _stream(self.device)))
        return th, ld

    def inverse_sampler(self, z, x) -> Tuple[torch.Tensor, bool]:
        """theta = inverse(z | x) through the persistent sampler's own arithmetic (sf_flow_inverse_from_noise_sampler:
        split-bf16 x3 hidden blocks for a MAF with H <= 64).  Returns (theta, split) -- ``split`` False when this
        flow's sampler is the all-fp32 path."""
        self._dev()
        z, x = _f32c(z, self.device), _f32c(x, self.device)
        B = z.shape[0]
        if z.shape != (B, self.spec.D) or x.shape != (B, self.spec.C):
            raise ValueError("shape mismatch")
        th = torch.empty_like(z)
        rc = self.lib.sf_flow_inverse_from_noise_sampler(self.handle, _ptr(z), _ptr(x), B, _ptr(th), _stream(self.device))
        if rc < 0:
            _lib.check(rc)
        self.last_sampler_rc = int(rc)   # 0 split-bf16 x3 pass functions, 2 the 16-row sampler's fp32 ones, 1 the generic fp32 path
        return th, rc == 0

    def set_sample_row_offset(self, row_offset: int) -> None:
        """The rows of the next sampling / acceptance calls are rows [row_offset, ...) of a larger catalogue: only the
        random streams see it (sf_flow_set_sample_row_offset)."""
        _lib.check(self.lib.sf_flow_set_sample_row_offset(self.handle, int(row_offset)))

    def train_path(self, B: int, want_dctx: bool = False) -> int:
        """0: one producer wave per 32-sample tile; 1 / 2: cooperative 16-row kernel, 4- / 8-wave workgroups."""
        return int(self.lib.sf_flow_train_path(self.handle, int(B), 1 if want_dctx else 0))

    def sample(self, x, S: int, lo=None, hi=None, seed: int = 0, max_attempts: Optional[int] = None,
               out: Optional[torch.Tensor] = None, return_counts: bool = False):
        """samples[M,S,D].  ``max_attempts`` None / 0: no ceiling -- a slot is retried while its galaxy still gets
        draws accepted (sf_flow_sample); a positive value is a hard ceiling per slot, after which the slot is a NaN row.
        ``out``: float32 on the flow's device, or FLOAT64 -- on the device or in pinned host memory (a ``pin_memory`` CPU
        tensor: the kernels then write the reference's host container directly, sf_flow_set_sample_output_f64)."""
        max_attempts = int(max_attempts or 0)
        self._dev()
        x = _f32c(x, self.device)
        M = x.shape[0]
        if x.dim() != 2 or x.shape[1] != self.spec.C:
            raise ValueError(f"x must be (M,{self.spec.C})")
        lo_t = None if lo is None else _f32c(lo, self.device)
        hi_t = None if hi is None else _f32c(hi, self.device)
        if out is None:
            out = torch.empty((M, S, self.spec.D), dtype=torch.float32, device=self.device)
        nd = torch.empty(M, dtype=torch.int32, device=self.device) if return_counts else None
        unfilled = C.c_int64(0)
        f64 = out.dtype == torch.float64
        if out.shape != (M, S, self.spec.D) or not out.is_contiguous() or out.dtype not in (torch.float32, torch.float64):
            raise ValueError("out must be a contiguous (M,S,D) float32 / float64 tensor")
        if out.device.type != "cuda" and not (out.device.type == "cpu" and out.is_pinned()):
            raise ValueError("out must live on the flow's device or in pinned host memory")
        if f64:
            _lib.check(self.lib.sf_flow_set_sample_output_f64(self.handle, 1))
        try:
            _lib.check(self.lib.sf_flow_sample(self.handle, _ptr(x), M, S, _ptr(lo_t), _ptr(hi_t),
                                               C.c_uint64(seed & (2 ** 64 - 1)), max_attempts, _ptr(out), _ptr(nd),
                                               C.byref(unfilled), _stream(self.device)))
        finally:
            if f64:
                self.lib.sf_flow_set_sample_output_f64(self.handle, 0)
        self.last_unfilled = int(unfilled.value)
        st4 = (C.c_float * 4)()
        _lib.check(self.lib.sf_flow_sample_stats(self.handle, st4))
        self.last_sample_stats = dict(dense_ms=st4[0], rounds=int(st4[1]), rejected_round0=int(st4[2]), evaluations=st4[3])
        return (out, nd) if return_counts else out

    def sample_slots(self, x, S: int, slots: torch.Tensor, out: torch.Tensor, lo=None, hi=None, seed: int = 0,
                     max_attempts: Optional[int] = None) -> int:
        """Fill the listed output slots (slot = g*S + p; int32 / uint32 device tensor) of ``out`` [M,S,D] in place
        (sf_flow_sample_slots); returns the number of slots left as NaN rows."""
        self._dev()
        x = _f32c(x, self.device)
        if x.dim() != 2 or x.shape[1] != self.spec.C:
            raise ValueError(f"x must be (M,{self.spec.C})")
        M = x.shape[0]
        if out.shape != (M, S, self.spec.D) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous float32 (M,S,D) tensor on the flow's device")
        if slots.dtype not in (torch.int32, torch.uint32) or not slots.is_contiguous() or slots.device != self.device:
            raise ValueError("slots must be a contiguous 32-bit integer tensor on the flow's device")
        lo_t = None if lo is None else _f32c(lo, self.device)
        hi_t = None if hi is None else _f32c(hi, self.device)
        unfilled = C.c_int64(0)
        _lib.check(self.lib.sf_flow_sample_slots(self.handle, _ptr(x), M, S, _ptr(slots), slots.numel(), _ptr(lo_t), _ptr(hi_t),
                                                 C.c_uint64(seed & (2 ** 64 - 1)), int(max_attempts or 0), _ptr(out),
                                                 C.byref(unfilled), _stream(self.device)))
        self.last_unfilled = int(unfilled.value)
        return self.last_unfilled

    def supports_f64_out(self) -> bool:
        """Whether sample() takes a float64 ``out`` (device or pinned host): every flow but the one-parameter / autoregressive NSF."""
        ok = self.lib.sf_flow_set_sample_output_f64(self.handle, 1) == 0
        self.lib.sf_flow_set_sample_output_f64(self.handle, 0)
        return ok

    def set_sample_time_limit(self, seconds: Optional[float]) -> None:
        """Wall-clock ceiling of later sample() / sample_slots() calls (None / 0 = none); see sf_flow_set_sample_time_limit."""
        _lib.check(self.lib.sf_flow_set_sample_time_limit(self.handle, C.c_double(float(seconds or 0.0))))

    def set_profiling(self, on: bool) -> None:
        """Bracket the training flow kernel of later loss_grad calls with HIP events (sf_flow_set_profiling)."""
        _lib.check(self.lib.sf_flow_set_profiling(self.handle, 1 if on else 0))

    def train_kernel_ms(self) -> float:
        """Duration of the flow kernel of the last profiled loss_grad call (sf_flow_train_stats)."""
        ms = C.c_float(0.0)
        _lib.check(self.lib.sf_flow_train_stats(self.handle, C.byref(ms)))
        return float(ms.value)

    def prepare_context(self, x) -> None:
        """Per-galaxy context table for the sample_round calls that follow with this same tensor ``x``."""
        self._dev()
        _lib.check(self.lib.sf_flow_prepare_context(self.handle, _ptr(x), x.shape[0], _stream(self.device)))

    def release_context(self) -> None:
        _lib.check(self.lib.sf_flow_release_context(self.handle))

    def sample_round(self, x, S, slots, slot_base, n_slots, attempt, seed, lo, hi, out, rejected, n_rejected,
                     n_drawn=None, stream_id: int = 0, attempts_per_slot: int = 1):
        self._dev()
        _lib.check(self.lib.sf_flow_sample_round(
            self.handle, _ptr(x), S, _ptr(slots), slot_base, n_slots, attempt, attempts_per_slot,
            C.c_uint64(seed & (2 ** 64 - 1)),
            stream_id, _ptr(lo), _ptr(hi), _ptr(out), _ptr(rejected), _ptr(n_rejected), _ptr(n_drawn),
            _stream(self.device)))

    def acceptance(self, x, n: int, lo, hi, seed: int = 0) -> torch.Tensor:
        self._dev()
        x = _f32c(x, self.device)
        lo_t, hi_t = _f32c(lo, self.device), _f32c(hi, self.device)
        cnt = torch.empty(x.shape[0], dtype=torch.int32, device=self.device)
        _lib.check(self.lib.sf_flow_acceptance(self.handle, _ptr(x), x.shape[0], n, _ptr(lo_t), _ptr(hi_t),
                                               C.c_uint64(seed & (2 ** 64 - 1)), _ptr(cnt), _stream(self.device)))
        return cnt.to(torch.float32) / float(n)

    def train_epoch(self, flat: torch.Tensor, theta: torch.Tensor, x: torch.Tensor, order: torch.Tensor,
                    n_batches: int, batch: int, grad_scale: float, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                    adam_desc, step0: int, max_norm: float, scratch: torch.Tensor, grad: torch.Tensor,
                    loss_sum: torch.Tensor, comm=None) -> None:
        """One epoch of (fused-gather loss_grad + clip + Adam) steps driven from C (sf_flow_train_epoch):
        ``order`` int64 [n_batches*batch] device rows into ``theta`` / ``x``; ``loss_sum`` float64 device scalar.
        ``comm`` (a ``synference_amd.comm.RcclComm``): data parallel -- every step's gradient is summed over the ranks by one
        RCCL all-reduce inside the call (sf_flow_train_epoch_dp); ``order`` then holds this rank's rows."""
        self._dev()
        for t in (flat, theta, x, exp_avg, exp_avg_sq, grad):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("train_epoch needs contiguous float32 tensors on the flow's device")
        if order.dtype != torch.int64 or order.numel() < n_batches * batch or loss_sum.dtype != torch.float64:
            raise ValueError("order must be int64 with n_batches*batch entries, loss_sum float64")
        _lib.check(self.lib.sf_flow_train_epoch_dp(
            self.handle, _ptr(flat), _ptr(theta), _ptr(x), _ptr(order), n_batches, batch, C.c_float(grad_scale),
            _ptr(exp_avg), _ptr(exp_avg_sq), C.byref(adam_desc), step0, C.c_float(max_norm), _ptr(scratch), _ptr(grad),
            _ptr(loss_sum), C.c_void_p(comm.handle if comm is not None else None), _stream(self.device)))

    def loss_grad_rows(self, flat: torch.Tensor, theta: torch.Tensor, x: torch.Tensor, rows: torch.Tensor,
                       grad_scale: float, grad_out: torch.Tensor, loss_sum: Optional[torch.Tensor] = None,
                       loss_out: Optional[torch.Tensor] = None) -> None:
        """loss_grad over the batch ``theta[rows], x[rows]`` with the gather fused into the kernel
        (sf_flow_loss_grad_rows); ``loss_sum``: optional float64 device scalar accumulating sum_b (-log p_b)."""
        self._dev()
        for t in (flat, theta, x, grad_out):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("loss_grad_rows needs contiguous float32 tensors on the flow's device")
        if rows.dtype != torch.int64 or not rows.is_contiguous():
            raise ValueError("rows must be a contiguous int64 tensor")
        _lib.check(self.lib.sf_flow_loss_grad_rows(self.handle, _ptr(flat), _ptr(theta), _ptr(x), _ptr(rows), rows.numel(),
                                                   C.c_float(grad_scale), None, _ptr(loss_out), _ptr(loss_sum),
                                                   _ptr(grad_out), None, _stream(self.device)))

    def loss_grad(self, flat: torch.Tensor, theta, x, grad_scale: float,
                  grad_out: Optional[torch.Tensor] = None, weights: Optional[torch.Tensor] = None,
                  dctx_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        self._dev()
        flat = _f32c(flat, self.device)
        theta, x = _f32c(theta, self.device), _f32c(x, self.device)
        B = theta.shape[0]
        loss = torch.empty(B, dtype=torch.float32, device=self.device)
        grad = grad_out if grad_out is not None else torch.empty_like(flat)
        wts = None if weights is None else _f32c(weights, self.device)
        _lib.check(self.lib.sf_flow_loss_grad_weighted(self.handle, _ptr(flat), _ptr(theta), _ptr(x), B,
                                                       C.c_float(grad_scale), _ptr(wts), _ptr(loss), _ptr(grad),
                                                       _ptr(dctx_out), _stream(self.device)))
        return loss, grad

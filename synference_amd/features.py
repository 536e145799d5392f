"""Feature transform on the device.

``flux_to_abmag`` is the AB branch of the reference's feature engineering
(ref: src/synference/sbi_runner.py:1698-1716: ``-2.5 log10(f_uJy) + 23.9``, negative fluxes set to
``norm_mag_limit``; 1927-1932: magnitudes fainter than the limit clipped to it; 1699-1702: errors
``2.5 sigma / (ln 10 f)``).  ``flux_to_asinh`` is the asinh-magnitude branch (ref: src/synference/utils.py:647-704,
used at sbi_runner.py:1718-1731) and ``scatter_depths`` the depth-noise augmentation of the library
(ref: sbi_runner.py:580-691, 0-D / 1-D depths).  ``pit_ranks`` (ref: sbi_runner.py:7153-7158) lives here too.
Normalisation to a reference band, extra feature columns and unit parsing stay host-side and out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def flux_to_abmag(flux_njy: torch.Tensor, err_njy: Optional[torch.Tensor] = None, norm_mag_limit: float = 50.0):
    """(N,C) fluxes in nJy on the GPU -> AB magnitudes (and magnitude errors when ``err_njy`` is given)."""
    if flux_njy.device.type != "cuda":
        raise RuntimeError("flux_to_abmag runs on the GPU (no CPU fallback)")
    def _aligned(t):  # the kernel moves float4: a contiguous view that starts off a 16-byte boundary is copied
        t = t.contiguous().float()
        return t if t.data_ptr() % 16 == 0 else t.clone()
    f = _aligned(flux_njy)
    e = None if err_njy is None else _aligned(err_njy)
    mag = torch.empty_like(f)
    mag_err = None if e is None else torch.empty_like(f)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(f.device).cuda_stream)
    _lib.check(_lib.load().sf_flux_to_abmag(p(f), p(e), f.numel(), C.c_float(norm_mag_limit), p(mag), p(mag_err), st))
    return mag if e is None else (mag, mag_err)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def flux_to_asinh(flux_njy: torch.Tensor, f_b_njy, err_njy: Optional[torch.Tensor] = None):
    """(N,C) fluxes in nJy on the GPU -> asinh magnitudes with softening ``f_b_njy`` (scalar or per band [C])."""
    if flux_njy.device.type != "cuda":
        raise RuntimeError("flux_to_asinh runs on the GPU (no CPU fallback)")
    f = flux_njy.contiguous().float()
    N, Cb = f.shape
    fb = torch.as_tensor(f_b_njy, dtype=torch.float32, device=f.device).reshape(-1)
    if fb.numel() == 1:
        fb = fb.expand(Cb)
    if fb.numel() != Cb:
        raise ValueError("Flux softening must match the number of filters.")
    fb = fb.contiguous()
    e = None if err_njy is None else err_njy.contiguous().float()
    mag = torch.empty_like(f)
    mag_err = None if e is None else torch.empty_like(f)
    _lib.check(_lib.load().sf_flux_to_asinh(_p(f), _p(e), N, Cb, _p(fb), _p(mag), _p(mag_err), _stream(f.device)))
    return mag if e is None else (mag, mag_err)


def scatter_depths(flux: torch.Tensor, depths, n_scatters: int = 5, depth_sigma: float = 5.0,
                   min_flux_pc_error: float = 0.0, seed: int = 0, return_errors: bool = False):
    """(N,C) library photometry -> (N*n_scatters, C) noisy copies, sigma = depths / depth_sigma per band
    (``depths`` scalar, [C], or [k, C] = k depth sets: as in the reference, one set is then drawn per band and
    scatter copy, from a generator seeded with ``seed``); row i*n_scatters + s is scatter s of row i."""
    if flux.device.type != "cuda":
        raise RuntimeError("scatter_depths runs on the GPU (no CPU fallback)")
    f = flux.contiguous().float()
    N, Cb = f.shape
    dep = torch.as_tensor(depths, dtype=torch.float32, device=f.device)
    if dep.dim() == 2:  # (k, C) depth sets: depths[idx[c, s], c] per band and scatter (sbi_runner.py:636-649)
        if dep.shape[1] != Cb:
            raise ValueError(f"Mismatch in dimensions: photometry has {Cb} bands but depths has {dep.shape[1]} columns")
        g = torch.Generator().manual_seed(int(seed) & 0x7FFFFFFF)
        idx = torch.randint(0, dep.shape[0], (n_scatters, Cb), generator=g).to(f.device)
        sg = dep.gather(0, idx) / float(depth_sigma)                       # [n_scatters, C]
    else:
        sg = dep.reshape(-1) / float(depth_sigma)
        if sg.numel() == 1:
            sg = sg.expand(Cb)
        if sg.numel() != Cb:
            raise ValueError(f"Mismatch in dimensions: photometry has {Cb} bands but depths has {sg.numel()} elements")
        sg = sg.reshape(1, Cb)
    sg = sg.contiguous()
    out = torch.empty((N * n_scatters, Cb), dtype=torch.float32, device=f.device)
    err = torch.empty_like(out) if return_errors else None
    _lib.check(_lib.load().sf_scatter_depths(_p(f), N, Cb, _p(sg), sg.shape[0], n_scatters, C.c_float(min_flux_pc_error),
                                             C.c_uint64(seed & (2 ** 64 - 1)), _p(out), _p(err), _stream(f.device)))
    return (out, err) if return_errors else out


def pit_ranks(samples: torch.Tensor, truth: torch.Tensor) -> torch.Tensor:
    """(N,S,D) draws and (N,D) truths on the GPU -> (N,D) fraction of finite draws below the truth."""
    if samples.device.type != "cuda":
        raise RuntimeError("pit_ranks runs on the GPU (no CPU fallback)")
    s = samples.contiguous().float()
    N, S, D = s.shape
    t = truth.to(s.device).contiguous().float().reshape(N, D)
    out = torch.empty((N, D), dtype=torch.float32, device=s.device)
    _lib.check(_lib.load().sf_pit_ranks(_p(s), _p(t), N, S, D, _p(out), _stream(s.device)))
    return out

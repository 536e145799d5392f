"""Feature transform on the device.

``flux_to_abmag`` is the AB branch of the reference's feature engineering
(ref: src/synference/sbi_runner.py:1698-1716: ``-2.5 log10(f_uJy) + 23.9``, negative fluxes set to
``norm_mag_limit``; 1927-1932: magnitudes fainter than the limit clipped to it; 1699-1702: errors
``2.5 sigma / (ln 10 f)``).  Everything else of ``create_feature_array*`` stays out of scope.
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
    f = flux_njy.contiguous().float()
    e = None if err_njy is None else err_njy.contiguous().float()
    mag = torch.empty_like(f)
    mag_err = None if e is None else torch.empty_like(f)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(f.device).cuda_stream)
    _lib.check(_lib.load().sf_flux_to_abmag(p(f), p(e), f.numel(), C.c_float(norm_mag_limit), p(mag), p(mag_err), st))
    return mag if e is None else (mag, mag_err)

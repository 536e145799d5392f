"""CPU restatement of the posterior wrappers around the flow.  TEST INFRASTRUCTURE ONLY.

Restates [UPSTREAM] sbi ``DirectPosterior`` / ``EnsemblePosterior`` /
``accept_reject_sample`` as they are driven by the reference at
src/synference/sbi_runner.py:6438-6442 (sample) and :7193-7196 (log_prob), with the
in-tree box predicate ``Interval.check`` (src/synference/custom_runner.py:982-987:
``low <= v <= high`` on every dimension).

Schedule difference, stated once: sbi draws whole batches per galaxy and keeps the
accepted rows until S are collected; here every output slot (galaxy g, draw p) is
its own rejection sampler on the counter stream (slot, attempt=0,1,2,...).  Both
deliver i.i.d. draws from the flow restricted to the prior box; the per-slot form
is independent of batching, which is what lets the HIP sampler and this oracle
agree draw for draw.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import flows, philox


def in_box(theta: np.ndarray, lo: Optional[np.ndarray], hi: Optional[np.ndarray]) -> np.ndarray:
    """custom_runner.py:986  ``(lower_bound <= value) & (value <= upper_bound)`` over all dims,
    plus finiteness (a NaN draw is never accepted)."""
    ok = np.isfinite(theta).all(-1)
    if lo is not None:
        ok &= ((theta >= lo) & (theta <= hi)).all(-1)
    return ok


def sample_slots(spec: flows.FlowSpec, flat: torch.Tensor, x: np.ndarray, slots: np.ndarray,
                 S: int, seed: int, lo=None, hi=None, max_attempts: int = 64, stream: int = 0,
                 dtype=torch.float32) -> Tuple[np.ndarray, np.ndarray]:
    """Draw one accepted sample for each slot id (slot = g*S + p).

    Returns (theta[len(slots), D], attempts_used[len(slots)]); rows that exhaust
    ``max_attempts`` are NaN (the reference's failure convention, sbi_runner.py:6458-6460).
    """
    slots = np.asarray(slots, dtype=np.uint64)
    out = np.full((len(slots), spec.D), np.nan, dtype=np.float64)
    used = np.zeros(len(slots), dtype=np.int64)
    pending = np.arange(len(slots))
    flat = flat.to(dtype)
    lo_ = None if lo is None else np.asarray(lo, dtype=np.float32)
    hi_ = None if hi is None else np.asarray(hi, dtype=np.float32)
    for attempt in range(max_attempts):
        if len(pending) == 0:
            break
        sl = slots[pending]
        g = (sl // np.uint64(S)).astype(np.int64)
        z = philox.normal(seed, sl, attempt, spec.D, stream=stream)
        with torch.no_grad():
            th, _ = flows.inverse_transform(spec, flat, torch.as_tensor(z).to(dtype),
                                            torch.as_tensor(np.asarray(x)[g]).to(dtype))
        th32 = th.to(torch.float32).numpy()
        ok = in_box(th32, lo_, hi_)
        out[pending[ok]] = th.numpy()[ok]
        used[pending] += 1
        pending = pending[~ok]
    return out, used


def sample(spec, flat, x, S, seed, lo=None, hi=None, max_attempts=64, dtype=torch.float32):
    """``posterior.sample((S,), x=x[g])`` for every row g -> (samples[M,S,D], n_drawn[M])."""
    M = len(x)
    th, used = sample_slots(spec, flat, x, np.arange(M * S, dtype=np.uint64), S, seed, lo, hi,
                            max_attempts, dtype=dtype)
    return th.reshape(M, S, spec.D), used.reshape(M, S).sum(1)


def acceptance(spec, flat, x, n: int, seed: int, lo, hi, dtype=torch.float32) -> np.ndarray:
    """Fraction of n unconstrained flow draws per row that fall in the prior box
    ([UPSTREAM] DirectPosterior.leakage_correction, num_rejection_samples=n); stream id 1."""
    M = len(x)
    sl = np.arange(M * n, dtype=np.uint64)
    z = philox.normal(seed, sl, 0, spec.D, stream=1)
    g = (sl // np.uint64(n)).astype(np.int64)
    with torch.no_grad():
        th, _ = flows.inverse_transform(spec, flat.to(dtype), torch.as_tensor(z).to(dtype),
                                        torch.as_tensor(np.asarray(x)[g]).to(dtype))
    ok = in_box(th.to(torch.float32).numpy(), np.asarray(lo, np.float32), np.asarray(hi, np.float32))
    return ok.reshape(M, n).mean(1)


def posterior_log_prob(spec, flat, theta, x, lo=None, hi=None, norm_posterior=False,
                       num_rejection_samples=10000, seed=0, dtype=torch.float32) -> np.ndarray:
    """[UPSTREAM] DirectPosterior.log_prob: raw flow density, -inf outside the prior
    support, minus log(acceptance) when ``norm_posterior``."""
    with torch.no_grad():
        lp = flows.log_prob(spec, flat.to(dtype), torch.as_tensor(np.asarray(theta)).to(dtype),
                            torch.as_tensor(np.asarray(x)).to(dtype)).double().numpy()
    if lo is not None:
        lp = np.where(in_box(np.asarray(theta, np.float32), np.asarray(lo, np.float32),
                             np.asarray(hi, np.float32)), lp, -np.inf)
        if norm_posterior:
            acc = acceptance(spec, flat, x, num_rejection_samples, seed, lo, hi, dtype)
            lp = lp - np.log(acc)
    return lp


# ---- ensemble ([UPSTREAM] sbi EnsemblePosterior; built in-tree at custom_runner.py:278-283) ----
def ensemble_counts(weights: Sequence[float], S: int, M: int, seed: int) -> np.ndarray:
    """Per-row multinomial split of the S draws over the members -> counts[M, E]."""
    w = np.asarray(weights, dtype=np.float64)
    w = w / w.sum()
    return np.random.default_rng(seed).multinomial(S, w, size=M)


def ensemble_sample(specs: List[flows.FlowSpec], flats: List[torch.Tensor], weights, x, S, seed,
                    lo=None, hi=None, dtype=torch.float32) -> np.ndarray:
    """Member e fills positions [cum_{e-1}, cum_e) of each row (member order, not shuffled)."""
    M = len(x)
    counts = ensemble_counts(weights, S, M, seed)
    cum = np.concatenate([np.zeros((M, 1), np.int64), np.cumsum(counts, 1)], 1)
    out = np.full((M, S, specs[0].D), np.nan)
    pos = np.arange(S)[None, :]
    for e, (sp, fl) in enumerate(zip(specs, flats)):
        mask = (pos >= cum[:, e:e + 1]) & (pos < cum[:, e + 1:e + 2])
        slots = np.flatnonzero(mask.reshape(-1)).astype(np.uint64)
        th, _ = sample_slots(sp, fl, x, slots, S, seed, lo, hi, dtype=dtype)
        out.reshape(M * S, -1)[slots.astype(np.int64)] = th
    return out


def ensemble_log_prob(specs, flats, weights, theta, x, lo=None, hi=None, dtype=torch.float32):
    """logsumexp_i(log w_i + lp_i)."""
    w = np.asarray(weights, dtype=np.float64)
    w = w / w.sum()
    lps = np.stack([posterior_log_prob(sp, fl, theta, x, lo, hi, dtype=dtype)
                    for sp, fl in zip(specs, flats)], 0)
    a = lps + np.log(w)[:, None]
    m = np.max(a, 0)
    with np.errstate(invalid="ignore"):
        r = m + np.log(np.exp(a - m).sum(0))
    return np.where(np.isfinite(m), r, -np.inf)

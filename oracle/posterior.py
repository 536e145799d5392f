"""CPU restatement of the posterior wrappers around the flow.  TEST INFRASTRUCTURE ONLY.

Restates [UPSTREAM] sbi ``DirectPosterior`` / ``EnsemblePosterior`` /
``accept_reject_sample`` as they are driven by the reference at
src/synference/sbi_runner.py:6438-6442 (sample) and :7193-7196 (log_prob), with the
in-tree box predicate ``Interval.check`` (src/synference/custom_runner.py:982-987:
``low <= v <= high`` on every dimension).

Two samplers live here.
``accept_reject_sample`` restates sbi's batch loop itself (first batch min(S, 10 000)
proposals, later batches resized to 1.5 x remaining / acceptance, accepted rows kept in
order until S are collected) on torch's own noise: the reference-shaped sampler that the
distributional tests (two-sample KS, PIT) compare the HIP sampler with.
``sample_slots`` / ``sample`` restate the product's schedule: every output slot
(galaxy g, draw p) is its own rejection sampler on the counter stream
(slot, attempt=0,1,2,...).  Both deliver i.i.d. draws from the flow restricted to the
prior box; the per-slot form is independent of batching, which is what lets the HIP
sampler and this oracle agree draw for draw.
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
                 S: int, seed: int, lo=None, hi=None, max_attempts: Optional[int] = None, stream: int = 0,
                 dtype=torch.float32, row_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Draw one accepted sample for each slot id (slot = g*S + p).  ``row_offset``: x holds rows [row_offset, ...) of a
    larger catalogue; the random streams are those of slot (row_offset + g)*S + p (sf_flow_set_sample_row_offset).

    ``max_attempts`` an integer: hard ceiling, rows that exhaust it are NaN (the reference's failure
    convention, sbi_runner.py:6458-6460).  ``None``: no ceiling, as in [UPSTREAM] accept_reject_sample
    -- with the product's progress rule (include/synference_hip.h, sf_flow_sample): the rule is looked at
    where a window ends (attempts 1024, 16384, 262144, ...) once the open slots have seen at least 1e5
    attempts (window length x S) since the last look; the open slots of a galaxy that got no draw
    accepted since then (counting from the 64th attempt) become NaN rows.
    Returns (theta[len(slots), D], attempts_used[len(slots)]).
    """
    slots = np.asarray(slots, dtype=np.uint64)
    out = np.full((len(slots), spec.D), np.nan, dtype=np.float64)
    used = np.zeros(len(slots), dtype=np.int64)
    pending = np.arange(len(slots))
    flat = flat.to(dtype)
    lo_ = None if lo is None else np.asarray(lo, dtype=np.float32)
    hi_ = None if hi is None else np.asarray(hi, dtype=np.float32)
    ceiling = int(max_attempts) if max_attempts else 1 << 30
    attempt, window_end, acc_from = 0, min(1024, ceiling), 64
    xs = np.asarray(x)
    progressed = set()
    while len(pending) and attempt < ceiling:
        while attempt < window_end and len(pending):
            sl = slots[pending]
            g = (sl // np.uint64(S)).astype(np.int64)
            z = philox.normal(seed, sl + np.uint64(row_offset * S), attempt, spec.D, stream=stream)
            with torch.no_grad():
                th, _ = flows.inverse_transform(spec, flat, torch.as_tensor(z).to(dtype),
                                                torch.as_tensor(xs[g]).to(dtype))
            ok = in_box(th.to(torch.float32).numpy(), lo_, hi_)
            out[pending[ok]] = th.numpy()[ok]
            used[pending] += 1
            if attempt >= 64:
                progressed.update(g[ok].tolist())
            pending = pending[~ok]
            attempt += 1
        if not max_attempts and len(pending) and (attempt - acc_from) * S >= 100000:
            g = (slots[pending] // np.uint64(S)).astype(np.int64)
            pending = pending[np.isin(g, list(progressed))]
            progressed, acc_from = set(), attempt
        window_end = min(ceiling, window_end * 16)
    return out, used


def accept_reject_sample(spec: flows.FlowSpec, flat: torch.Tensor, x_row: np.ndarray, S: int, lo, hi,
                         generator: torch.Generator, max_sampling_batch_size: int = 10_000,
                         warn_acceptance: float = 0.01, dtype=torch.float32):
    """[UPSTREAM] sbi ``accept_reject_sample`` as ``DirectPosterior.sample((S,), x=x_row)`` runs it (reached from
    ref: sbi_runner.py:6442; box predicate custom_runner.py:982-987): propose ``flow.sample(b, context=x_row)``
    in batches, keep the rows inside the prior box, first batch min(S, 10 000), then
    ``min(10 000, max(int(1.5 * remaining / acceptance), 100))``, until S are kept; the first S kept rows are
    returned.  Returns (samples[S, D] float64, acceptance_rate, low_acceptance_warned)."""
    flat = flat.to(dtype)
    lo_ = np.asarray(lo, dtype=np.float32)
    hi_ = np.asarray(hi, dtype=np.float32)
    xr = torch.as_tensor(np.asarray(x_row)).to(dtype).reshape(1, -1)
    kept, n_kept, n_total, warned = [], 0, 0, False
    bs = min(S, max_sampling_batch_size)
    while n_kept < S:
        z = torch.randn(bs, spec.D, generator=generator, dtype=torch.float64).to(dtype)
        with torch.no_grad():
            th, _ = flows.inverse_transform(spec, flat, z, xr.expand(bs, -1))
        ok = in_box(th.to(torch.float32).numpy(), lo_, hi_)
        kept.append(th.double().numpy()[ok])
        n_kept += int(ok.sum())
        n_total += bs
        rate = n_kept / n_total
        bs = min(max_sampling_batch_size, max(int(1.5 * (S - n_kept) / max(rate, 1e-12)), 100))
        if n_total > 1000 and rate < warn_acceptance:
            warned = True
    return np.concatenate(kept, 0)[:S], n_kept / n_total, warned


def accept_reject_sample_batched(spec: flows.FlowSpec, flat: torch.Tensor, x: np.ndarray, S: int, lo, hi,
                                 generator: torch.Generator, dtype=torch.float32, max_rounds: int = 200):
    """The same rejection sampler as ``accept_reject_sample`` for a whole catalogue at once (what a batched CPU
    implementation of the reference's loop would do; the CPU baseline of bench.py that scales over cores): every round
    proposes one draw for every still-empty slot of every galaxy in ONE batched inverse pass (torch.randn noise), keeps
    the draws inside the box, repeats.  Returns (samples[M, S, D] float32 -- NaN where ``max_rounds`` ran out --,
    proposals drawn)."""
    flat = flat.to(dtype)
    lo_ = torch.as_tensor(np.asarray(lo, dtype=np.float32))
    hi_ = torch.as_tensor(np.asarray(hi, dtype=np.float32))
    xs = torch.as_tensor(np.asarray(x)).to(dtype)
    M = xs.shape[0]
    out = torch.full((M * S, spec.D), float("nan"), dtype=torch.float32)
    pending = torch.arange(M * S)
    drawn = 0
    for _ in range(max_rounds):
        if pending.numel() == 0:
            break
        z = torch.randn(pending.numel(), spec.D, generator=generator, dtype=dtype)
        with torch.no_grad():
            th, _ = flows.inverse_transform(spec, flat, z, xs[pending // S])
        th32 = th.to(torch.float32)
        ok = ((th32 >= lo_) & (th32 <= hi_)).all(-1)
        out[pending[ok]] = th32[ok]
        drawn += int(pending.numel())
        pending = pending[~ok]
    return out.reshape(M, S, spec.D).numpy(), drawn


def sample(spec, flat, x, S, seed, lo=None, hi=None, max_attempts=None, dtype=torch.float32, row_offset=0):
    """``posterior.sample((S,), x=x[g])`` for every row g -> (samples[M,S,D], n_drawn[M])."""
    M = len(x)
    th, used = sample_slots(spec, flat, x, np.arange(M * S, dtype=np.uint64), S, seed, lo, hi,
                            max_attempts, dtype=dtype, row_offset=row_offset)
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

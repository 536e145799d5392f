"""Posterior protocol of the reference on the HIP engine.

``SBI_Fitter`` consumes: ``.sample((S,), x=)`` (ref: sbi_runner.py:6442 through
ili ``DirectSampler``), ``.log_prob(x=, theta=)`` (7196) with optional
``leakage_correction_params`` (custom_runner.py:466-473), ``.sample_batched((S,), x=X)`` ->
``(S,N,D)`` (custom_runner.py:489-493), indexable ``.posteriors`` / ``.weights`` / ``.name`` /
``.signatures`` (custom_runner.py:278-285).  ``FlowPosterior`` restates [UPSTREAM] sbi
``DirectPosterior`` and ``EnsemblePosterior`` restates sbi ``EnsemblePosterior``;
the catalogue-wide calls (``sample_catalogue`` / ``log_prob_catalogue``) are what the reference's
per-galaxy Python loops collapse into.

Rejection schedule: every output slot (galaxy g, draw p) is its own rejection sampler on the
Philox stream (slot, attempt); accepted draws are i.i.d. from the flow restricted to the prior
box exactly as with sbi's batch loop, but the result does not depend on batching.
"""
from __future__ import annotations

import logging
from typing import List, Optional, Sequence

import numpy as np
import torch

from .estimator import FlowEstimator
from .priors import CustomIndependentUniform

logger = logging.getLogger("synference_amd")

_MAX_SLOTS_PER_CALL = 1 << 27  # keeps slot ids in 32 bits and scratch lists modest


def _as_x2d(x, C, device):
    x = torch.as_tensor(x, dtype=torch.float32, device=device)
    if x.dim() == 1:
        x = x[None, :]
    if x.dim() != 2 or x.shape[1] != C:
        raise ValueError(f"x must have trailing dimension {C}, got {tuple(x.shape)}")
    return x.contiguous()


class FlowPosterior:
    """q(theta | x) restricted to the prior support ([UPSTREAM] sbi DirectPosterior)."""

    def __init__(self, posterior_estimator: FlowEstimator, prior: Optional[CustomIndependentUniform] = None,
                 max_sampling_attempts: Optional[int] = None, seed: int = 0):
        """``max_sampling_attempts``: None (default) = no ceiling, like [UPSTREAM] accept_reject_sample, which draws
        until S samples are kept: a slot is retried for as long as its galaxy still gets draws accepted (see
        sf_flow_sample); an integer is a hard ceiling on the attempts per output slot, after which the slot is a NaN
        row (the reference's failure convention, sbi_runner.py:6458-6460)."""
        self.posterior_estimator = posterior_estimator
        self.prior = prior
        self.max_sampling_attempts = None if not max_sampling_attempts else int(max_sampling_attempts)
        self._seed = int(seed)
        self._calls = 0
        self.name = ""
        self.signatures = None
        self.last_acceptance = None

    # ---- helpers ----------------------------------------------------------------------------
    @property
    def device(self):
        return self.posterior_estimator.flat.device

    @property
    def spec(self):
        return self.posterior_estimator.spec

    def to(self, device):
        self.posterior_estimator.to(device)
        if self.prior is not None:
            self.prior = self.prior.to(device)
        return self

    def _embed(self, X):
        """raw features -> the (N, C) context the flow consumes (embedding nets run once per galaxy)."""
        est = self.posterior_estimator
        if not getattr(est, "has_embedding", False):
            return _as_x2d(X, self.spec.C, self.device)
        X = torch.as_tensor(X, dtype=torch.float32, device=self.device)
        if X.dim() == 1:
            X = X[None, :]
        with torch.no_grad():
            return est.embed(X).contiguous()

    def _box(self):
        if self.prior is None:
            return None, None
        return self.prior.low.to(self.device), self.prior.high.to(self.device)

    def _next_seed(self, seed):
        if seed is not None:
            return int(seed)
        self._calls += 1
        return (self._seed * 0x9E3779B97F4A7C15 + self._calls) & (2 ** 63 - 1)

    # ---- catalogue-wide fast paths ----------------------------------------------------------
    def sample_catalogue(self, X, num_samples: int, seed: Optional[int] = None, return_counts=False,
                         timeout_seconds: Optional[float] = None, row_offset: int = 0, out: Optional[torch.Tensor] = None):
        """(N,S,D) float32 device tensor of accepted draws for every row of X (NaN rows on failure).
        ``out``: optional result tensor to fill instead -- float32 on the device, or float64 on the device / in pinned host
        memory (HipFlow.sample: the kernels then write the host container of SBI_Fitter.sample_posterior directly).
        ``timeout_seconds``: wall-clock ceiling of the call (the reference's ``timeout_seconds_per_test`` x objects).
        ``row_offset``: X holds rows [row_offset, row_offset + N) of a larger catalogue (a rank's shard): with the same
        seed the draws are those of a single call over the whole catalogue, whatever the chunking or the sharding."""
        est = self.posterior_estimator
        X = self._embed(X)
        est._sync_params()
        lo, hi = self._box()
        seed = self._next_seed(seed)
        N, S = X.shape[0], int(num_samples)
        # the ceiling is the CALL's: a chunk gets what is left of it (never less than a second, so that a late chunk still
        # runs its first launch and reports NaN rows instead of raising)
        import time as _time
        t_call = _time.monotonic()
        if out is None:
            out = torch.empty((N, S, self.spec.D), dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (N, S, self.spec.D):
            raise ValueError("out must be (N, S, D)")
        counts = torch.empty(N, dtype=torch.int32, device=self.device)
        rows_per = max(1, _MAX_SLOTS_PER_CALL // max(S, 1))
        # keep the per-galaxy context table of a chunk within 2 GiB so that it is always built
        desc = est.flow.describe()
        per_gal = 4 * int(desc.get("ctab_floats_per_galaxy", 0))
        if per_gal > 0:
            rows_per = max(1, min(rows_per, (2 << 30) // per_gal))
        # shape cliffs are not silent: a lampe-backend flow outside the register-tile sampler's shapes draws ~5 x slower
        if desc.get("sampler_tiles16", 1) == 0 and self.spec.D > 1 and N * S >= 100000 and not getattr(est, "_warned_sampler", False):
            est._warned_sampler = True
            sp = self.spec
            logger.warning(f"{sp.kind} D={sp.D} C={sp.C} H={sp.H} T={sp.T}: this shape samples on the 64-sample LDS kernel (k_ar_sample), not on "
                           "the 16-candidate register-tile kernels (2 <= D <= 8, at most 32 hidden units per parameter, D + C <= 32)")
        unfilled = 0
        try:
            for r0 in range(0, N, rows_per):
                r1 = min(N, r0 + rows_per)
                # slot ids restart per chunk; the random streams are keyed by the row's position in the whole catalogue
                est.flow.set_sample_row_offset(int(row_offset) + r0)
                est.flow.set_sample_time_limit(None if timeout_seconds is None
                                               else max(1.0, float(timeout_seconds) - (_time.monotonic() - t_call)))
                o, c = est.flow.sample(X[r0:r1], S, lo, hi, seed=seed, max_attempts=self.max_sampling_attempts,
                                       out=out[r0:r1], return_counts=True)
                counts[r0:r1] = c
                unfilled += est.flow.last_unfilled
        finally:
            est.flow.set_sample_row_offset(0)
        # one read-back for everything the host wants to know about the call: mean / worst acceptance, galaxies below 1 %
        if N:   # (ONE small read-back -- N attempt counters -- and numpy, instead of six device kernels and a read-back)
            acc_g = S / np.maximum(counts.cpu().numpy().astype(np.float64), 1.0)
            summ = [float(acc_g.mean()), float(acc_g.min()), float((acc_g < 0.01).sum())]
        self.last_acceptance = summ[0] if N else None
        self.last_unfilled = unfilled
        if unfilled:
            why = (f"after {self.max_sampling_attempts} attempts" if self.max_sampling_attempts
                   else "(their galaxies' acceptance rate is zero to sampling precision)")
            logger.error(f"{unfilled} posterior draws could not be placed inside the prior support {why}; "
                         "those rows are NaN.")
        if N:
            # [UPSTREAM] accept_reject_sample warns per observation once its acceptance rate drops below 1 %
            # ("... It may take a long time to collect the remaining samples"); the in-tree diagnostics of
            # custom_runner.py:1131-1186 name the offending parameters -- CustomIndependentUniform.acceptance_report
            n_low = int(summ[2])
            self.last_low_acceptance = n_low
            if n_low:
                logger.warning(f"Only {summ[1] * 100:.3f}% of the proposed samples were accepted for the worst of "
                               f"{n_low} observation(s) below 1 %: their posterior mass is largely outside the prior support.")
        return (out, counts) if return_counts else out

    def acceptance_rows(self, ux, num_rejection_samples: int, seed: int, row_offset: int = 0):
        """Leakage-correction acceptance rate of every row of ``ux`` (rows [row_offset, ...) of a larger list of distinct
        contexts): chunked, streams keyed by the row's position in the whole list."""
        est = self.posterior_estimator
        lo, hi = self._box()
        n_rej = int(num_rejection_samples)
        # one acceptance launch numbers its draws with 32 bits (and its context table is capped at 2 GiB)
        rows_per = max(1, min(_MAX_SLOTS_PER_CALL // max(n_rej, 1), 1 << 20))
        per_gal = 4 * int(est.flow.describe().get("ctab_floats_per_galaxy", 0))
        if per_gal > 0:
            rows_per = max(1, min(rows_per, (2 << 30) // per_gal))
        acc = torch.empty(ux.shape[0], dtype=torch.float32, device=self.device)
        try:
            for r0 in range(0, ux.shape[0], rows_per):
                r1 = min(ux.shape[0], r0 + rows_per)
                est.flow.set_sample_row_offset(int(row_offset) + r0)
                acc[r0:r1] = est.flow.acceptance(ux[r0:r1], n_rej, lo, hi, seed=seed)
        finally:
            est.flow.set_sample_row_offset(0)
        return acc

    def log_prob_catalogue(self, theta, X, norm_posterior: bool = True, num_rejection_samples: int = 10000,
                           seed: Optional[int] = None, acc_rows=None):
        """[UPSTREAM] DirectPosterior.log_prob for aligned rows: raw density, -inf outside the prior
        support, minus log(acceptance(x)) when ``norm_posterior``.  ``acc_rows``: acceptance rate of
        every row, computed by the caller (rank-sharded evaluation: the distinct contexts are split over the ranks)."""
        est = self.posterior_estimator
        theta = torch.as_tensor(theta, dtype=torch.float32, device=self.device)
        if theta.dim() == 1:
            theta = theta[None, :]
        X = self._embed(X)
        if X.shape[0] == 1 and theta.shape[0] > 1:
            X = X.expand(theta.shape[0], -1).contiguous()
        est._sync_params()
        lp = est.flow.log_prob(theta, X)
        lo, hi = self._box()
        if lo is not None:
            inside = ((theta >= lo) & (theta <= hi)).all(-1)
            lp = torch.where(inside, lp, torch.full_like(lp, float("-inf")))
            if norm_posterior and acc_rows is not None:
                lp = lp - torch.log(torch.as_tensor(acc_rows, dtype=torch.float32, device=lp.device).clamp_min(1e-30))
            elif norm_posterior:
                ux, inv = torch.unique(X, dim=0, return_inverse=True)
                acc = self.acceptance_rows(ux, num_rejection_samples, self._next_seed(seed))
                lp = lp - torch.log(acc.clamp_min(1e-30))[inv]
        return lp

    # ---- sbi surface ------------------------------------------------------------------------
    def sample(self, sample_shape=(1,), x=None, show_progress_bars=False, seed: Optional[int] = None, **_):
        S = int(np.prod(sample_shape)) if len(tuple(sample_shape)) else 1
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        x = x[None, :] if x.dim() == 1 else x
        if x.shape[0] != 1:
            raise ValueError("sample() takes ONE observation; use sample_batched() for a catalogue")
        out = self.sample_catalogue(x, S, seed)[0]
        return out.reshape(*tuple(sample_shape), self.spec.D)

    def sample_batched(self, sample_shape=(1,), x=None, show_progress_bars=False, seed: Optional[int] = None, **_):
        S = int(np.prod(sample_shape))
        out = self.sample_catalogue(x, S, seed)  # (N,S,D)
        return out.permute(1, 0, 2).contiguous()  # sbi convention (S,N,D)

    def log_prob(self, theta, x=None, norm_posterior: bool = True, leakage_correction_params=None, **_):
        p = dict(leakage_correction_params or {})
        return self.log_prob_catalogue(theta, x, norm_posterior, p.get("num_rejection_samples", 10000))

    def log_prob_batched(self, theta, x, norm_posterior: bool = True, **_):
        """theta (S,N,D), x (N,C) -> (S,N)."""
        theta = torch.as_tensor(theta, dtype=torch.float32, device=self.device)
        S, N, D = theta.shape
        X = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        lp = self.log_prob_catalogue(theta.reshape(S * N, D), X.repeat(S, 1), norm_posterior)
        return lp.reshape(S, N)

    def potential_fn(self, theta, x):
        return self.log_prob_catalogue(theta, x, norm_posterior=False)


class EnsemblePosterior:
    """Weighted mixture of posteriors ([UPSTREAM] sbi EnsemblePosterior; built in the reference at
    custom_runner.py:278-283, weights from validation log-probs in ili's runner)."""

    def __init__(self, posteriors: Sequence[FlowPosterior], weights=None, theta_transform=None, seed: int = 0):
        self.posteriors: List[FlowPosterior] = list(posteriors)
        n = len(self.posteriors)
        w = torch.ones(n) / n if weights is None else torch.as_tensor(weights, dtype=torch.float32).detach().cpu()
        self._weights = w / w.sum()
        self.theta_transform = theta_transform
        self.name = ""
        self.signatures = None
        self._seed = int(seed)
        self._calls = 0

    def __len__(self):
        return len(self.posteriors)

    @property
    def weights(self):
        return self._weights

    @property
    def device(self):
        return self.posteriors[0].device

    @property
    def prior(self):
        return self.posteriors[0].prior

    def to(self, device):
        for p in self.posteriors:
            p.to(device)
        return self

    def _next_seed(self, seed):
        if seed is not None:
            return int(seed)
        self._calls += 1
        return (self._seed * 0x9E3779B97F4A7C15 + 0x51ED27 + self._calls) & (2 ** 63 - 1)

    def sample_catalogue(self, X, num_samples: int, seed: Optional[int] = None, timeout_seconds: Optional[float] = None,
                         row_offset: int = 0, out: Optional[torch.Tensor] = None):
        """Per row: multinomial(weights, S) split; member e fills positions [cum_{e-1}, cum_e).  ``row_offset``: X holds
        rows [row_offset, ...) of a larger catalogue (see FlowPosterior.sample_catalogue)."""
        if len(self.posteriors) == 1:
            return self.posteriors[0].sample_catalogue(X, num_samples, self._next_seed(seed), timeout_seconds=timeout_seconds,
                                                       row_offset=row_offset, out=out)
        if out is not None:
            raise ValueError("a caller-provided result tensor is only taken by a one-member posterior")
        for p in self.posteriors:
            p.posterior_estimator.flow.set_sample_time_limit(timeout_seconds)
        p0 = self.posteriors[0]
        dev, D, S = p0.device, p0.spec.D, int(num_samples)
        X = torch.as_tensor(X, dtype=torch.float32, device=dev)
        X = X[None, :] if X.dim() == 1 else X
        N = X.shape[0]
        seed = self._next_seed(seed)
        out = torch.full((N, S, D), float("nan"), dtype=torch.float32, device=dev)
        w = self._weights.double().numpy()
        # (the generator fills rows in order: the split of a row depends on its position in the WHOLE catalogue only)
        counts = np.random.default_rng(seed & 0xFFFFFFFF).multinomial(S, w / w.sum(), size=int(row_offset) + N)[int(row_offset):]
        cum = np.concatenate([np.zeros((N, 1), np.int64), np.cumsum(counts, 1)], 1)
        rows_per = max(1, (_MAX_SLOTS_PER_CALL // 4) // max(S, 1))
        pos = torch.arange(S, device=dev)[None, :]
        for r0 in range(0, N, rows_per):
            r1 = min(N, r0 + rows_per)
            cum_d = torch.as_tensor(cum[r0:r1], device=dev)
            for e, post in enumerate(self.posteriors):
                mask = (pos >= cum_d[:, e:e + 1]) & (pos < cum_d[:, e + 1:e + 2])
                slots = torch.nonzero(mask.reshape(-1)).reshape(-1).to(torch.int32)  # bit pattern == uint32
                flow = post.posterior_estimator.flow
                try:
                    flow.set_sample_row_offset(int(row_offset) + r0)
                    _sample_slot_list(post, post._embed(X[r0:r1]), S, slots, seed, out[r0:r1])
                finally:
                    flow.set_sample_row_offset(0)
        return out

    def sample(self, sample_shape=(1,), x=None, show_progress_bars=False, seed: Optional[int] = None, **_):
        S = int(np.prod(sample_shape)) if len(tuple(sample_shape)) else 1
        x = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        x = x[None, :] if x.dim() == 1 else x
        if x.shape[0] != 1:
            raise ValueError("sample() takes ONE observation; use sample_batched() for a catalogue")
        return self.sample_catalogue(x, S, seed)[0].reshape(*tuple(sample_shape), -1)

    def sample_batched(self, sample_shape=(1,), x=None, show_progress_bars=False, seed: Optional[int] = None, **_):
        return self.sample_catalogue(x, int(np.prod(sample_shape)), seed).permute(1, 0, 2).contiguous()

    def log_prob_catalogue(self, theta, X, norm_posterior: bool = True, num_rejection_samples: int = 10000, acc=None):
        """``acc``: optional list (one entry per member) of precomputed acceptance rates per row (rank-sharded callers)."""
        lps = torch.stack([p.log_prob_catalogue(theta, X, norm_posterior, num_rejection_samples,
                                                **({} if acc is None else {"acc_rows": acc[i]}))
                           for i, p in enumerate(self.posteriors)], 0)
        logw = torch.log(self._weights.to(lps.device))[:, None]
        return torch.logsumexp(lps + logw, dim=0)

    def log_prob(self, theta, x=None, norm_posterior: bool = True, leakage_correction_params=None, **_):
        p = dict(leakage_correction_params or {})
        return self.log_prob_catalogue(theta, x, norm_posterior, p.get("num_rejection_samples", 10000))

    def potential_fn(self, theta, x):
        return self.log_prob_catalogue(theta, x, norm_posterior=False)


def device_quantiles(samples: torch.Tensor, quantiles) -> torch.Tensor:
    """(N,S,D) float32 device draws -> (N,D,Q) quantiles on the device (sf_quantiles; numpy 'linear' rule,
    NaN draws ignored).  Counterpart of the host pass at ref: sbi_runner.py:3270-3282."""
    import ctypes as C
    from . import _lib
    if samples.device.type != "cuda":
        raise RuntimeError("device_quantiles needs the draws on the GPU")
    samples = samples.contiguous().float()
    N, S, D = samples.shape
    q = torch.as_tensor(np.asarray(quantiles, dtype=np.float32), device=samples.device)
    out = torch.empty((N, D, q.numel()), dtype=torch.float32, device=samples.device)
    st = C.c_void_p(torch.cuda.current_stream(samples.device).cuda_stream)
    _lib.check(_lib.load().sf_quantiles(C.c_void_p(samples.data_ptr()), N, S, D, C.c_void_p(q.data_ptr()), q.numel(),
                                        C.c_void_p(out.data_ptr()), st))
    return out


def _sample_slot_list(post: FlowPosterior, X, S: int, slots: torch.Tensor, seed: int, out: torch.Tensor):
    """One member's share of an ensemble draw: the persistent sampler over an explicit slot list
    (sf_flow_sample_slots; ensemble members own disjoint slot sets of every row)."""
    n = int(slots.numel())
    if n == 0:
        return
    est = post.posterior_estimator
    est._sync_params()
    lo, hi = post._box()
    unfilled = est.flow.sample_slots(X, S, slots.contiguous(), out, lo, hi, seed=seed,
                                     max_attempts=post.max_sampling_attempts)
    if unfilled:
        logger.error(f"{unfilled} posterior draws could not be placed inside the prior support; those rows are NaN.")


# ---- rank-sharded catalogue evaluation (contiguous row blocks per GPU, weights replicated, no data-path
# collective; the blocks are gathered afterwards) -------------------------------------------------------------------------
def dist_world():
    """(rank, world) of the initialised default process group, else (0, 1)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, world: int):
    """Contiguous row blocks [b[r], b[r+1]) of n rows over ``world`` ranks."""
    return [(r * n) // world for r in range(world + 1)]


def all_gather_rows(local: torch.Tensor, bounds):
    """Every rank contributes rows [bounds[r], bounds[r+1]) (``local``: that block); returns the whole array on every
    rank.  Blocks are padded to the largest one for the collective; NCCL/RCCL takes device tensors, other backends
    (gloo: CPU tests, single-device rehearsals) are staged through the host."""
    import torch.distributed as dist
    rank, world = dist_world()
    if world == 1:
        return local
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    mx = max(sizes)
    nccl = dist.get_backend() == "nccl"
    src = local if nccl else local.cpu()
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=src.device)
    pad[: sizes[rank]] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.cat([parts[r][: sizes[r]] for r in range(world)], 0)
    return out.to(local.device)


_GATHER_OK = {}


def _backend_has_gather(device) -> bool:
    """Whether the process group's backend implements ``gather``: probed ONCE per backend with a one-element collective that
    every rank enters together, and the outcome agreed on by an all-reduce(MIN) -- so that no rank can take the all-gather
    fallback while another sits in a gather (an error inside a LATER collective is an error, never a reason to switch)."""
    import torch.distributed as dist
    key = dist.get_backend()
    if key not in _GATHER_OK:
        rank, world = dist_world()
        ok = 1
        try:
            t = torch.zeros(1, device=device)
            dist.gather(t, [torch.empty_like(t) for _ in range(world)] if rank == 0 else None, dst=0)
        except (RuntimeError, NotImplementedError):
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        _GATHER_OK[key] = bool(int(flag.item()))
    return _GATHER_OK[key]


def gather_rows(local: torch.Tensor, bounds, dst: int = 0):
    """Rank ``dst`` receives the whole array (rows [bounds[r], bounds[r+1]) from rank r) and returns it; every other rank
    returns None.  For the big outputs -- (N, S, D) draws: 3.2 GB for BASELINE configs[4] -- of which one copy on one rank is
    what the caller wants (per-rank slices, gathered once), where all_gather_rows would move world x the array
    over xGMI to hold world identical copies."""
    import torch.distributed as dist
    rank, world = dist_world()
    if world == 1:
        return local
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    mx = max(sizes)
    nccl = dist.get_backend() == "nccl"
    src = local if nccl else local.cpu()
    if sizes[rank] == mx:
        pad = src.contiguous()
    else:
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=src.device)
        pad[: sizes[rank]] = src
    if _backend_has_gather(pad.device):
        parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, parts, dst=dst)
    else:   # (the all-gather every backend has does the job at world x the traffic)
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
    if rank != dst:
        return None
    out = torch.cat([parts[r][: sizes[r]] for r in range(world)], 0)
    return out.to(local.device)


def broadcast_seed(seed):
    """The same seed on every rank (rank 0's: a call with seed=None draws one from the posterior's own counter)."""
    import torch.distributed as dist
    rank, world = dist_world()
    if world == 1:
        return seed
    box = [seed]
    dist.broadcast_object_list(box, src=0)
    return box[0]

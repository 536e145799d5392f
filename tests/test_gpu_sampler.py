"""Rejection-sampling semantics of the HIP sampler on the GPU.

* the per-slot schedule without an attempt ceiling against the oracle's restatement of it, draw for draw, at a
  prior box that accepts only ~1-3 % of the proposals;
* the give-up rule (a galaxy whose acceptance is zero) next to galaxies that fill normally;
* distributional equivalence with the reference-shaped batch sampler ([UPSTREAM] sbi accept_reject_sample as
  reached from ref: src/synference/sbi_runner.py:6442, restated in oracle/posterior.py on torch's own noise):
  two-sample Kolmogorov-Smirnov per (galaxy, parameter) and the reference's PIT statistic
  (ref: sbi_runner.py:7128-7160);
* the chunked leakage correction of log_prob over more distinct rows than one acceptance launch takes.
"""
import numpy as np
import pytest
import torch
from scipy import stats

from cases import make_case
from oracle import posterior as OP
from synference_amd.engine import HipFlow

pytestmark = pytest.mark.gpu


def _flow(spec, flat):
    f = HipFlow(spec, "cuda:0")
    f.set_params(torch.as_tensor(flat))
    return f


def _quantile_box(ospec, flat, x, qlo, qhi, n=600, seed=99):
    free, _ = OP.sample(ospec, torch.as_tensor(flat), x, n, seed, dtype=torch.float32)
    free = free.reshape(-1, ospec.D)
    return (np.quantile(free, qlo, axis=0).astype(np.float32), np.quantile(free, qhi, axis=0).astype(np.float32))


@pytest.mark.parametrize("name,qlo,qhi", [("maf_small", 0.44, 0.56), ("nsf_nb1", 0.30, 0.70), ("maf_cfg1", 0.25, 0.75)])
def test_uncapped_sampler_fills_a_low_acceptance_box_draw_for_draw(name, qlo, qhi):
    """No attempt ceiling (the default): every slot is retried until it is filled -- hundreds of attempts per
    slot here -- and still equals the oracle's sequential per-slot rejection sampler draw for draw."""
    ospec, spec, flat, theta, x = make_case(name, B=3, spread=0.2)
    lo, hi = _quantile_box(ospec, flat, x, qlo, qhi)
    S, seed = 96, 77
    f = _flow(spec, flat)
    got, nd = f.sample(x, S, lo, hi, seed=seed, return_counts=True)
    got, nd = got.cpu().double().numpy(), nd.cpu().numpy()
    assert f.last_unfilled == 0 and np.isfinite(got).all()
    assert ((got >= lo) & (got <= hi)).all()
    ref, rnd = OP.sample(ospec, torch.as_tensor(flat), x, S, seed, lo, hi, dtype=torch.float32)
    assert np.isfinite(ref).all()
    acc = S * len(x) / rnd.sum()
    assert acc < 0.06, acc                                   # the box really is hard
    assert rnd.max() > 64 * 2                                # some slot went past the first attempt window
    err = np.abs((got - ref) / (hi - lo).astype(np.float64)).max(-1)
    assert (err > 5e-4).mean() < 0.03, ((err > 5e-4).mean(), err.max())   # boundary accept/reject flips only
    assert np.abs(nd - rnd).sum() <= 0.05 * rnd.sum()


def test_uncapped_sampler_gives_up_only_on_dead_galaxies():
    """A galaxy that gets no draw accepted from its 64th attempt to the end of a window (here: a NaN context row, every
    draw is non-finite) ends as NaN rows; its neighbours are filled exactly as if it were not there."""
    ospec, spec, flat, theta, x = make_case("maf_small", B=5, spread=0.2)
    lo, hi = _quantile_box(ospec, flat, x, 0.10, 0.90)
    x = x.copy()
    x[2, :] = np.nan
    S, seed = 120, 5
    f = _flow(spec, flat)
    got = f.sample(x, S, lo, hi, seed=seed).cpu().double().numpy()
    assert f.last_unfilled == S
    assert np.isnan(got[2]).all()
    live = [0, 1, 3, 4]
    assert np.isfinite(got[live]).all()
    assert f.last_sample_stats["rounds"] == 1                # one launch: 960 x 120 attempts without a draw -> dropped at 1024
    ref, _ = OP.sample(ospec, torch.as_tensor(flat), x, S, seed, lo, hi, dtype=torch.float32)
    assert np.isnan(ref[2]).all()
    err = np.abs((got[live] - ref[live]) / (hi - lo).astype(np.float64)).max(-1)
    assert (err > 5e-4).mean() < 0.02
    # everything unreachable: nothing is filled and the call still returns
    got = f.sample(x[:2], 40, np.full(spec.D, 1e6, np.float32), np.full(spec.D, 2e6, np.float32), seed=1)
    assert f.last_unfilled == 80 and torch.isnan(got).all()
    assert f.last_sample_stats["rounds"] >= 2                # 40 slots: not enough evidence at 1024, given up at 16384
    # a caller-set ceiling still means what it says
    got = f.sample(x[:2], 40, lo, hi, seed=1, max_attempts=1)
    assert 0 < f.last_unfilled < 80
    assert int(torch.isnan(got).all(-1).sum()) == f.last_unfilled


@pytest.mark.parametrize("name", ["maf_cfg1", "nsf_cfg3"])
def test_hip_sampler_matches_reference_shaped_batch_sampler_in_distribution(name):
    """The HIP sampler's per-slot schedule and sbi's batch accept/reject loop must sample the SAME distribution
    (the flow restricted to the prior box).  Different noise, so the check is statistical: two-sample KS per
    (galaxy, parameter) with a Bonferroni-corrected threshold, and the reference's PIT values of both samplers
    against each other."""
    ospec, spec, flat, theta, x = make_case(name, B=6, spread=0.3)
    lo, hi = _quantile_box(ospec, flat, x, 0.08, 0.92)
    S = 4000
    f = _flow(spec, flat)
    hip = f.sample(x, S, lo, hi, seed=31).cpu().double().numpy()
    assert f.last_unfilled == 0
    gen = torch.Generator().manual_seed(123)
    ref = np.stack([OP.accept_reject_sample(ospec, torch.as_tensor(flat), x[g], S, lo, hi, gen)[0] for g in range(len(x))])
    assert ref.shape == hip.shape
    n_tests = len(x) * spec.D
    alpha = 1e-3 / n_tests                                   # family-wise 1e-3
    pvals = np.array([[stats.ks_2samp(hip[g, :, d], ref[g, :, d]).pvalue for d in range(spec.D)] for g in range(len(x))])
    assert pvals.min() > alpha, (pvals.min(), np.unravel_index(pvals.argmin(), pvals.shape))
    assert np.median(pvals) > 0.05                           # and no systematic shift hiding under the threshold
    # PIT as the reference defines it (sbi_runner.py:7153-7158): mean over draws and parameters of [sample < truth]
    truth = ref[:, :200, :]                                  # "truths" drawn from the reference-shaped sampler itself
    pit_hip = np.array([[(hip[g] < truth[g, k]).mean() for k in range(200)] for g in range(len(x))]).ravel()
    pit_ref = np.array([[(ref[g, 200:] < truth[g, k]).mean() for k in range(200)] for g in range(len(x))]).ravel()
    assert stats.ks_2samp(pit_hip, pit_ref).pvalue > 1e-3
    # with the truths drawn from the sampled distribution, per-parameter ranks are uniform (calibration)
    ranks = np.array([[(hip[g, :, d] < truth[g, k, d]).mean() for k in range(200)] for g in range(len(x)) for d in range(spec.D)])
    assert stats.kstest(ranks.ravel(), "uniform").pvalue > 1e-4


def test_log_prob_leakage_correction_is_chunked_over_many_distinct_rows():
    """norm_posterior=True estimates the acceptance of EVERY distinct x with num_rejection_samples draws: 5e5 rows x
    10 000 = 5e9 draws do not fit one acceptance launch (32-bit item ids) and must be chunked."""
    from synference_amd.estimator import FlowEstimator
    from synference_amd.posterior import FlowPosterior
    from synference_amd.priors import CustomIndependentUniform
    ospec, spec, flat, theta, x = make_case("maf_small", B=64, spread=0.2)
    lo, hi = _quantile_box(ospec, flat, x, 0.02, 0.98)
    est = FlowEstimator(spec, torch.as_tensor(flat), device="cuda:0").to("cuda:0")
    post = FlowPosterior(est, CustomIndependentUniform(lo, hi, [f"p{i}" for i in range(spec.D)], device="cuda:0"))
    N = 500_000
    rng = np.random.default_rng(3)
    X = (rng.normal(size=(N, spec.C)) * np.asarray(ospec.x_std) + np.asarray(ospec.x_mean)).astype(np.float32)
    TH = np.tile(0.5 * (lo + hi), (N, 1)).astype(np.float32)
    lp = post.log_prob_catalogue(torch.as_tensor(TH), torch.as_tensor(X), norm_posterior=True, num_rejection_samples=10000,
                                 seed=9)
    raw = post.log_prob_catalogue(torch.as_tensor(TH), torch.as_tensor(X), norm_posterior=False)
    assert lp.shape == (N,) and torch.isfinite(lp).all()
    corr = (lp - raw).cpu().double().numpy()
    assert (corr >= -1e-6).all() and corr.max() < 2.0        # acceptance in (0, 1]: the correction only raises log p
    # spot-check a few rows against the oracle's acceptance estimate (same Philox stream id, own seed -> statistical)
    idx = [0, 1234, N - 1]
    racc = OP.acceptance(ospec, torch.as_tensor(flat), X[idx], 10000, 9, lo, hi)
    assert np.abs(np.exp(-corr[idx]) - racc).max() < 0.03


@pytest.mark.parametrize("name,q", [("maf_small", 0.48), ("nsf_nb1", 0.385)])
def test_deep_tail_slots_beyond_the_first_window_match_the_oracle(name, q):
    """Acceptance around 1e-3: most slots outlive the persistent launch's attempt window (1024 on the 16-row MAF kernel,
    256 on the 32-row kernels) and are finished by the chip-wide find + resolve launches -- still the LOWEST accepted
    attempt of every slot, so still the oracle's draws."""
    ospec, spec, flat, theta, x = make_case(name, B=2, spread=0.2)
    lo, hi = _quantile_box(ospec, flat, x, q, 1.0 - q, n=4000)
    S, seed = 16, 13
    f = _flow(spec, flat)
    got, nd = f.sample(x, S, lo, hi, seed=seed, return_counts=True)
    got, nd = got.cpu().double().numpy(), nd.cpu().numpy()
    assert f.last_unfilled == 0 and np.isfinite(got).all() and ((got >= lo) & (got <= hi)).all()
    assert f.last_sample_stats["rounds"] >= 2
    ref, rnd = OP.sample(ospec, torch.as_tensor(flat), x, S, seed, lo, hi, dtype=torch.float32)
    assert rnd.max() > 16 * 150                              # mean attempts per slot in the hundreds
    err = np.abs((got - ref) / (hi - lo).astype(np.float64)).max(-1)
    assert (err > 5e-4).mean() < 0.1, ((err > 5e-4).mean(), err.max())
    assert np.abs(nd - rnd).sum() <= 0.1 * rnd.sum()


def test_deep_tail_find_and_resolve_launches_equal_the_oracle_draw_for_draw():
    """A box so tight (acceptance ~ 2e-3) that most slots use up the persistent launch's 1 024 attempts: they are filled
    by the find / resolve launches of the deep tail (k_maf_find16s on this shape: the sampler's own split-bf16
    arithmetic) and must still be the oracle's draws -- the LOWEST accepted attempt of every slot."""
    ospec, spec, flat, theta, x = make_case("maf_cfg1", B=2, spread=0.2)
    lo, hi = _quantile_box(ospec, flat, x, 0.36, 0.64, n=2000)
    S, seed = 24, 5
    f = _flow(spec, flat)
    got, nd = f.sample(x, S, lo, hi, seed=seed, return_counts=True)
    got, nd = got.cpu().double().numpy(), nd.cpu().numpy()
    assert f.last_unfilled == 0 and np.isfinite(got).all()
    assert f.last_sample_stats["rounds"] >= 3               # persistent launch + at least one find / resolve pair
    ref, rnd = OP.sample(ospec, torch.as_tensor(flat), x, S, seed, lo, hi, dtype=torch.float32)
    acc = S * len(x) / rnd.sum()
    assert acc < 5e-3, acc
    assert (rnd > 1024 * S * 0.5).any()                     # a galaxy whose slots mostly went past the first window
    err = np.abs((got - ref) / (hi - lo).astype(np.float64)).max(-1)
    assert (err > 5e-4).mean() <= 0.05, ((err > 5e-4).mean(), err.max())   # boundary accept/reject flips only
    assert np.abs(nd - rnd).sum() <= 0.08 * rnd.sum()

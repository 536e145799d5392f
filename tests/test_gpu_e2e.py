"""End-to-end through the reference's API surface on the GPU (SBI_Fitter -> HIPRunner -> posterior)."""
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fitted(tmp_path_factory):
    from synference_amd import SBI_Fitter
    from synference_amd.synthetic import make_catalogue
    x, theta, names = make_catalogue(4000, 10, 5, seed=1)
    f = SBI_Fitter("e2e", names, [f"F{i}" for i in range(10)], feature_array=x, parameter_array=theta)
    out = tmp_path_factory.mktemp("models")
    post, stats = f.run_single_sbi(model_type="maf", hidden_features=50, num_transforms=5, n_nets=2,
                                   training_batch_size=256, learning_rate=2e-3, stop_after_epochs=3,
                                   max_num_epochs=12, random_seed=3, out_dir=str(out), verbose=False,
                                   name_append="t", evaluate_model=False)
    return f, post, stats, out / "e2e"     # (out_dir gets the fitter's name appended: sbi_runner.py:4541)


def test_training_reduces_loss_and_fills_stats(fitted):
    f, post, stats, out = fitted
    assert len(post) == 2 and len(stats) == 2
    for s in stats:
        for k in ("training_log_probs", "validation_log_probs", "best_validation_log_prob", "epochs_trained",
                  "training_loss", "validation_loss", "best_validation_loss", "converged"):
            assert k in s
        assert s["training_loss"][-1] < s["training_loss"][0] - 0.5
        assert s["training_log_probs"][0] == -s["training_loss"][0]
    assert abs(float(post.weights.sum()) - 1.0) < 1e-6
    assert (out / "e2e_t_posterior.pkl").exists() and (out / "e2e_t_summary.json").exists()


def test_sample_posterior_contract(fitted):
    f, post, stats, _ = fitted
    X = f._X_test[:50]
    s = f.sample_posterior(X, num_samples=200, seed=5)
    assert s.shape == (50, 200, 5) and s.dtype == np.float64
    assert np.isfinite(s).all()
    lo, hi = f._prior.low.numpy(), f._prior.high.numpy()
    assert ((s >= lo) & (s <= hi)).all()
    one = f.sample_posterior(X[0], num_samples=64, seed=5)
    assert one.shape == (64, 5)
    # the trained posterior must be informative: posterior mean closer to truth than the prior mean
    mean = s.mean(1)
    y = f._y_test[:50]
    prior_mean = (lo + hi) / 2
    assert np.mean((mean[:, 0] - y[:, 0]) ** 2) < 0.7 * np.mean((prior_mean[0] - y[:, 0]) ** 2)


def test_log_prob_contract_and_ensemble_mixture(fitted):
    f, post, stats, _ = fitted
    X, y = f._X_test[:40], f._y_test[:40]
    raw = f.log_prob(X, y, norm_posterior=False)
    assert raw.shape == (40,) and raw.dtype == np.float64 and np.isfinite(raw).all()
    # mixture rule: logsumexp_i(log w_i + lp_i)
    lps = np.stack([p.log_prob_catalogue(torch.as_tensor(y, dtype=torch.float32), torch.as_tensor(X),
                                         norm_posterior=False).double().cpu().numpy() for p in post.posteriors])
    w = post.weights.double().numpy()
    ref = np.log((np.exp(lps - lps.max(0)) * w[:, None]).sum(0)) + lps.max(0)
    assert np.abs(raw - ref).max() < 1e-4
    norm = f.log_prob(X, y, norm_posterior=True, num_rejection_samples=2000)
    assert (norm >= raw - 1e-5).all()      # acceptance <= 1 can only raise the normalised density
    outside = y.copy()
    outside[:, 0] = 1e3
    assert np.isneginf(f.log_prob(X, outside)).all()


def test_fit_catalogue_quantiles_and_nan_rows(fitted):
    import pandas as pd
    f, post, stats, _ = fitted
    X = f._X_test[:20].copy()
    X[3, 2] = np.nan
    df = pd.DataFrame(X, columns=f.feature_names)
    table = f.fit_catalogue(df, num_samples=300, seed=1)
    for p in f.simple_fitted_parameter_names:
        for q in (16, 50, 84):
            assert f"{p}_{q}" in table.columns
        v = table[[f"{p}_16", f"{p}_50", f"{p}_84"]].to_numpy()
        assert np.isnan(v[3]).all()
        ok = np.delete(v, 3, axis=0)
        assert (ok[:, 0] <= ok[:, 1]).all() and (ok[:, 1] <= ok[:, 2]).all()


def test_posterior_pickle_roundtrip(fitted):
    f, post, stats, out = fitted
    with open(out / "e2e_t_posterior.pkl", "rb") as fh:
        p2 = pickle.load(fh)
    p2.to("cuda:0")
    X = f._X_test[:5]
    a = post.sample_catalogue(torch.as_tensor(X), 50, seed=9).cpu().numpy()
    b = p2.sample_catalogue(torch.as_tensor(X), 50, seed=9).cpu().numpy()
    assert np.array_equal(a, b)


def test_saved_state_reloads_into_a_fresh_fitter(fitted):
    """ref: sbi_runner.py:693-830 (save_state), 7401-7633 (load_model_from_pkl): the directory written by
    run_single_sbi(save_model=True) is everything a later session needs -- posterior, stats, names, prior, arrays, split."""
    from synference_amd import SBI_Fitter
    f, post, stats, out = fitted
    assert (out / "e2e_t_params.pkl").exists() and (out / "e2e_t_summary.json").exists()
    g = SBI_Fitter("e2e", list(f.parameter_names))
    p2, stats2, params = g.load_model_from_pkl(str(out))                      # the directory holds exactly one model
    assert params["n_nets"] == len(post.posteriors) and params["train_args"]["training_batch_size"] > 0
    assert list(g.feature_names) == list(f.feature_names) and g.fitted_parameter_names == list(f.fitted_parameter_names)
    assert np.array_equal(g.feature_array, f.feature_array) and np.array_equal(g._X_test, f._X_test)
    assert np.allclose(g._prior.low.cpu(), f._prior.low.cpu()) and len(stats2) == len(stats) + 1   # + the scalar summary
    X = f._X_test[:5]
    a = f.sample_posterior(X, num_samples=40, seed=3)
    b = g.sample_posterior(X, num_samples=40, seed=3)
    assert np.array_equal(a, b)
    t1 = f.fit_catalogue(X, num_samples=100, seed=4, append_to_input=False)
    t2 = g.fit_catalogue(X, num_samples=100, seed=4, append_to_input=False)
    assert np.allclose(t1.to_numpy(float), t2.to_numpy(float), equal_nan=True)
    with pytest.raises(ValueError, match="does not exist"):
        g.load_model_from_pkl(str(out / "nope"))
    (out / "second_posterior.pkl").write_bytes(b"x")
    with pytest.raises(ValueError, match="Multiple parameter files"):
        g.load_model_from_pkl(str(out))
    (out / "second_posterior.pkl").unlink()


def test_unsupported_requests_fail_loudly(fitted):
    f, *_ = fitted
    with pytest.raises(ValueError, match="not on the HIP path"):
        f.run_single_sbi()                                        # the reference's default model_type is "mdn"
    with pytest.raises(ValueError):
        f.run_single_sbi(backend="pydelfi", model_type="maf")
    with pytest.raises(ValueError, match="online"):
        f.run_single_sbi(model_type="maf", learning_type="online")
    with pytest.warns(UserWarning, match="unknown keyword"):
        with pytest.raises(ValueError):
            f.run_single_sbi(model_type="mdn", not_a_reference_argument=1)
    with pytest.raises(ValueError):
        f.sample_posterior(f._X_test[:2], sample_method="emcee")


def test_autograd_estimator_matches_direct_loss_grad(fitted):
    """nn.Module surface: mean(-log_prob).backward() == the direct uniform-weight HIP backward."""
    f, post, _, _ = fitted
    est = post.posteriors[0].posterior_estimator
    th = torch.as_tensor(f._y_test[:100], dtype=torch.float32, device="cuda")
    x = torch.as_tensor(f._X_test[:100], device="cuda")
    est.zero_grad(set_to_none=True)
    losses = est.loss(th, x)
    (losses * torch.linspace(0.5, 1.5, 100, device="cuda")).mean().backward()
    g_auto = est.flat.grad.clone()
    _, g_ref = est.flow.loss_grad(est.flat.detach(), th, x, 1.0 / 100, weights=torch.linspace(0.5, 1.5, 100))
    assert (g_auto - g_ref).abs().max() <= 1e-5 * g_ref.abs().max() + 1e-8


def test_nsf_fit_and_ensemble_sampling_match_oracle(tmp_path):
    """NSF through the same surface, then a 2-member ensemble (one MAF-free: two NSFs) sampled on the GPU
    against the oracle's ensemble rule (multinomial split per row, member order)."""
    from cases import make_case
    from oracle import posterior as OP
    from synference_amd import SBI_Fitter
    from synference_amd.estimator import FlowEstimator
    from synference_amd.posterior import EnsemblePosterior, FlowPosterior
    from synference_amd.priors import CustomIndependentUniform
    from synference_amd.synthetic import make_catalogue
    x, theta, names = make_catalogue(3000, 20, 8, seed=2)
    f = SBI_Fitter("nsf", names, [f"F{i}" for i in range(20)], feature_array=x, parameter_array=theta)
    post, stats = f.run_single_sbi(model_type="nsf", hidden_features=50, num_transforms=3,
                                   additional_model_args={"num_bins": 8}, training_batch_size=256,
                                   learning_rate=2e-3, stop_after_epochs=2, max_num_epochs=6, random_seed=1,
                                   save_model=False, verbose=False)
    assert stats[0]["training_loss"][-1] < stats[0]["training_loss"][0] - 0.5
    s = f.sample_posterior(f._X_test[:10], num_samples=100, seed=3)
    assert s.shape == (10, 100, 8) and np.isfinite(s).all()
    # ---- ensemble vs oracle, draw for draw
    o1, s1, fl1, _, xx = make_case("nsf_odd", seed=0, B=5, spread=0.2)
    o2, s2, fl2, _, _ = make_case("nsf_odd", seed=0, B=5, spread=0.25)
    free, _ = OP.sample(o1, torch.as_tensor(fl1), xx, 300, 99, dtype=torch.float32)
    lo = np.quantile(free.reshape(-1, s1.D), 0.03, axis=0).astype(np.float32)
    hi = np.quantile(free.reshape(-1, s1.D), 0.97, axis=0).astype(np.float32)
    prior = CustomIndependentUniform(lo, hi, device="cuda")
    ests = [FlowEstimator(s1, torch.as_tensor(fl1)).to("cuda"), FlowEstimator(s2, torch.as_tensor(fl2)).to("cuda")]
    ens = EnsemblePosterior([FlowPosterior(e, prior) for e in ests], weights=[0.3, 0.7])
    got = ens.sample_catalogue(torch.as_tensor(xx), 128, seed=17).cpu().double().numpy()
    ref = OP.ensemble_sample([o1, o2], [torch.as_tensor(fl1), torch.as_tensor(fl2)], [0.3, 0.7], xx, 128, 17, lo, hi)
    err = np.abs((got - ref) / (hi - lo)).max(-1)
    assert np.isfinite(got).all() and (err > 5e-4).mean() < 1e-2, (err > 5e-4).mean()
    lp = ens.log_prob_catalogue(torch.as_tensor(ref[:, 0].astype(np.float32)), torch.as_tensor(xx),
                                norm_posterior=False).cpu().double().numpy()
    rlp = OP.ensemble_log_prob([o1, o2], [torch.as_tensor(fl1), torch.as_tensor(fl2)], [0.3, 0.7],
                               ref[:, 0].astype(np.float32), xx, lo, hi)
    assert np.abs(lp - rlp).max() < 2e-4


def test_lampe_backend_fits_the_autoregressive_nsf(tmp_path):
    """backend="lampe" (ref: sbi_runner.py:5123-5125 -> ili.utils.load_nde_lampe -> zuko.flows.NSF) through the fitter: the net
    factory builds the autoregressive NSF (8 bins, bound 5), training lowers the loss, the posterior samples inside the prior
    box, and the saved model reloads to the same density."""
    from synference_amd import SBI_Fitter
    from synference_amd.estimator import load_nde_hip
    from synference_amd.synthetic import make_catalogue
    x, theta, names = make_catalogue(3000, 10, 5, seed=4)
    net = load_nde_hip("NPE", model="nsf", backend="lampe", hidden_features=32, num_transforms=3, device="cuda")
    est = net(batch_theta=theta[:500], batch_x=x[:500])
    assert est.spec.kind == "nsf_ar" and est.spec.K == 8 and est.spec.tail_bound == 5.0 and est.flow.train_path(256) == 5
    with pytest.raises(ValueError, match="not on the HIP path"):
        load_nde_hip("NPE", model="mdn", backend="lampe")
    # model "maf" of the same backend = zuko.flows.MAF (affine univariate map on the same hyper-network)
    est_m = load_nde_hip("NPE", model="maf", backend="lampe", hidden_features=32, num_transforms=3, device="cuda")(batch_theta=theta[:500],
                                                                                                                 batch_x=x[:500])
    assert est_m.spec.kind == "maf_ar" and est_m.flow.train_path(256) == 5
    f = SBI_Fitter("lampe_nsf", names, [f"F{i}" for i in range(10)], feature_array=x, parameter_array=theta)
    post, stats = f.run_single_sbi(backend="lampe", model_type="nsf", hidden_features=32, num_transforms=3, training_batch_size=256,
                                   learning_rate=2e-3, stop_after_epochs=2, max_num_epochs=6, random_seed=1,
                                   save_model=True, out_dir=str(tmp_path), verbose=False)
    assert stats[0]["training_loss"][-1] < stats[0]["training_loss"][0] - 0.5
    s = f.sample_posterior(f._X_test[:10], num_samples=100, seed=3)
    lo, hi = theta.min(0), theta.max(0)
    assert s.shape == (10, 100, 5) and np.isfinite(s).all() and (s >= lo - 1e-6).all() and (s <= hi + 1e-6).all()
    lp = f.log_prob(f._X_test[:10], f._y_test[:10], norm_posterior=False)
    assert np.isfinite(np.asarray(lp)).all()
    fm = SBI_Fitter("lampe_maf", names, [f"F{i}" for i in range(10)], feature_array=x, parameter_array=theta)
    postm, statsm = fm.run_single_sbi(backend="lampe", model_type="maf", hidden_features=32, num_transforms=3, training_batch_size=256,
                                      learning_rate=2e-3, stop_after_epochs=2, max_num_epochs=6, random_seed=1, save_model=False,
                                      verbose=False, plot=False, evaluate_model=False)
    assert postm.posteriors[0].spec.kind == "maf_ar" and statsm[0]["training_loss"][-1] < statsm[0]["training_loss"][0] - 0.5
    sm = fm.sample_posterior(fm._X_test[:10], num_samples=100, seed=3)
    assert sm.shape == (10, 100, 5) and np.isfinite(sm).all() and (sm >= lo - 1e-6).all() and (sm <= hi + 1e-6).all()


def test_lampe_reference_example_ensemble_fits(tmp_path):
    """The reference's own backend="lampe" example (examples/sbi/scripts/basic_model.py:31-41): an ensemble of three NSFs with
    hidden_features [180, 150, 120] and num_transforms [16, 10, 6] -- its defining arguments verbatim, the run bounded by
    max_num_epochs.  Round 4 refused the 180-unit member (three hidden buffers of LDS); the training sweep now runs on two."""
    from synference_amd import SBI_Fitter
    from synference_amd.synthetic import make_catalogue
    x, theta, names = make_catalogue(3000, 12, 7, seed=6)
    f = SBI_Fitter("basic", names, [f"F{i}" for i in range(12)], feature_array=x, parameter_array=theta)
    post, stats = f.run_single_sbi(n_nets=3, backend="lampe", engine="NPE", name_append="_ensemble_lampe_nsf", stop_after_epochs=15,
                                   hidden_features=[180, 150, 120], learning_rate=0.0004, num_transforms=[16, 10, 6], model_type="nsf",
                                   max_num_epochs=3, training_batch_size=256, out_dir=str(tmp_path), verbose=False, plot=False,
                                   evaluate_model=False, random_seed=2)
    assert len(post.posteriors) == 3 and len(stats) == 3
    shapes = [(p.spec.kind, p.spec.H, p.spec.T, p.spec.K) for p in post.posteriors]
    assert shapes == [("nsf_ar", 180, 16, 8), ("nsf_ar", 150, 10, 8), ("nsf_ar", 120, 6, 8)]
    for s in stats:
        assert np.isfinite(s["training_loss"]).all() and s["training_loss"][-1] < s["training_loss"][0]
    s = f.sample_posterior(f._X_test[:6], num_samples=200, seed=3)
    lo, hi = theta.min(0), theta.max(0)
    assert s.shape == (6, 200, 7) and np.isfinite(s).all() and (s >= lo - 1e-6).all() and (s <= hi + 1e-6).all()
    lp = f.log_prob(f._X_test[:6], f._y_test[:6], norm_posterior=False)
    assert np.isfinite(np.asarray(lp)).all()
    assert (tmp_path / "basic" / "basic__ensemble_lampe_nsf_posterior.pkl").exists()


@pytest.mark.parametrize("name", ["maf_cfg1", "nsf_cfg3", "maf_span6", "maf_wide"])
def test_sampler_writes_the_float64_host_container_directly(name, monkeypatch):
    """sf_flow_set_sample_output_f64: the sampling kernels widen every accepted draw in its store, into a float64 array on the
    device or in PINNED HOST memory (the reference's container, sbi_runner.py:6436) -- same draws, bit for bit, as fp32 output
    widened afterwards, NaN rows included; and SBI_Fitter.sample_posterior takes that path for a one-member posterior."""
    import sys
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from cases import make_case
    from synference_amd.engine import HipFlow
    ospec, spec, flat, theta, x = make_case(name, B=40, spread=0.2)
    f = HipFlow(spec, "cuda:0")
    f.set_params(torch.as_tensor(flat))
    assert f.supports_f64_out()
    X = torch.as_tensor(x, dtype=torch.float32, device="cuda:0")
    free = f.sample(X[:4], 300, seed=1).cpu().numpy().reshape(-1, spec.D)
    lo, hi = np.quantile(free, 0.04, axis=0).astype(np.float32), np.quantile(free, 0.96, axis=0).astype(np.float32)
    S = 333
    ref = f.sample(X, S, lo, hi, seed=9).double().cpu()
    d64 = torch.empty((40, S, spec.D), dtype=torch.float64, device="cuda:0")
    f.sample(X, S, lo, hi, seed=9, out=d64)
    h64 = torch.full((40, S, spec.D), -7.0, dtype=torch.float64).pin_memory()
    f.sample(X, S, lo, hi, seed=9, out=h64)
    torch.cuda.synchronize()
    assert torch.equal(d64.cpu(), ref) and torch.equal(h64, ref)
    # a hard attempt ceiling leaves NaN rows: written as float64 NaNs too
    tight_hi = (lo + 0.02 * (hi - lo)).astype(np.float32)
    r32 = f.sample(X, 64, lo, tight_hi, seed=3, max_attempts=4).double().cpu()
    h2 = torch.zeros((40, 64, spec.D), dtype=torch.float64).pin_memory()
    f.sample(X, 64, lo, tight_hi, seed=3, max_attempts=4, out=h2)
    torch.cuda.synchronize()
    assert torch.isnan(r32).any() and torch.equal(torch.nan_to_num(h2, nan=-1.0), torch.nan_to_num(r32, nan=-1.0))
    with pytest.raises(ValueError):
        f.sample(X, S, lo, hi, seed=9, out=torch.empty((40, S, spec.D), dtype=torch.float64))      # unpinned host memory


def test_sample_posterior_direct_host_output_equals_the_staged_copy(fitted, monkeypatch):
    f, post, stats, _ = fitted
    from synference_amd.posterior import EnsemblePosterior
    one = EnsemblePosterior([post.posteriors[0]], weights=[1.0])
    X = f._X_test[:300]
    a = f.sample_posterior(X, num_samples=500, seed=21, posteriors=one)            # direct: pinned float64 written by the kernels
    monkeypatch.setenv("SF_API_DIRECT", "0")
    b = f.sample_posterior(X, num_samples=500, seed=21, posteriors=one)            # fp32 on the device, copied + widened
    assert a.dtype == np.float64 and a.shape == (300, 500, 5) and np.array_equal(a, b, equal_nan=True)
    monkeypatch.delenv("SF_API_DIRECT")
    keep = a.copy()
    c = f.sample_posterior(X, num_samples=500, seed=22, posteriors=one)            # a held result is never recycled
    assert np.array_equal(a, keep) and not np.array_equal(a, c)


def test_native_host_handover_is_exact():
    """sf_copy_to_host_f64 (csrc/sf_hostio.hip): float64 host copy of device fp32, bit for bit, for sizes that do not divide into
    pieces or vector widths, a destination that is not 32-byte aligned, and a second call reusing the ring."""
    from synference_amd import hostio
    g = torch.Generator(device="cuda").manual_seed(0)
    for n, shift in ((1, 0), (7, 1), (1234567, 3), (2 * 1024 * 1024 + 5, 0), (3 * 1048576, 2)):
        src = torch.randn(n, device="cuda", generator=g)
        src[::97] = float("nan")
        dst = np.empty(n + shift, np.float64)[shift:]
        hostio.to_host_f64(src, out=dst)
        assert np.array_equal(dst, src.double().cpu().numpy(), equal_nan=True)
    cube = torch.randn(50, 40, 5, device="cuda", generator=g)
    pend = hostio.to_host_f64(cube, wait=False)
    assert np.array_equal(pend.result(), cube.double().cpu().numpy())
    assert hostio.to_host_f64(cube[:0]).shape == (0, 40, 5)


def test_device_quantiles_match_numpy(fitted):
    from synference_amd.posterior import device_quantiles
    rng = np.random.default_rng(0)
    for S in (1, 7, 1000, 1024, 3000):
        a = rng.normal(size=(6, S, 3)).astype(np.float32)
        if S >= 7:
            a[1, :3, 0] = np.nan          # a few NaN draws: ignored
            a[2, :, 1] = np.nan           # all NaN -> NaN
        got = device_quantiles(torch.as_tensor(a).cuda(), [0.16, 0.5, 0.84]).cpu().numpy()
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = np.nanquantile(a.astype(np.float64), [0.16, 0.5, 0.84], axis=1).transpose(1, 2, 0)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.nanmax(np.abs(got - ref)) < 2e-6 * max(1.0, np.nanmax(np.abs(ref)))
    # the fitter's two routes agree
    f = fitted[0]
    import pandas as pd
    df = pd.DataFrame(f._X_test[:12], columns=f.feature_names)
    t_dev = f.fit_catalogue(df, num_samples=500, seed=4, device_quantiles=True)
    t_host = f.fit_catalogue(df, num_samples=500, seed=4, device_quantiles=False)
    for c in t_dev.columns:
        if c.endswith(("_16", "_50", "_84")):
            assert np.allclose(t_dev[c].to_numpy(), t_host[c].to_numpy(), rtol=1e-5, atol=1e-5), c


def test_flux_to_abmag_matches_reference_formula():
    from synference_amd.features import flux_to_abmag
    rng = np.random.default_rng(0)
    f = (10 ** rng.uniform(-3, 5, size=(1001, 7))).astype(np.float32)
    f[3, 2] = -5.0; f[4, 1] = 0.0; f[5, 0] = np.nan; f[6, 6] = 1e-30
    e = (0.1 * np.abs(f) + 1).astype(np.float32)
    mag, merr = flux_to_abmag(torch.as_tensor(f).cuda(), torch.as_tensor(e).cuda(), 50.0)
    with np.errstate(all="ignore"):
        ref = -2.5 * np.log10(f.astype(np.float64) / 1000.0) + 23.9      # sbi_runner.py:1705
        ref[f < 0] = 50.0                                                  # :1706, :1714
        ref[ref > 50.0] = 50.0                                             # :1932 (f == 0 -> +inf -> limit)
        rerr = 2.5 * e.astype(np.float64) / (np.log(10) * f.astype(np.float64))
    got = mag.cpu().double().numpy()
    assert np.isnan(got[5, 0]) and np.isnan(ref[5, 0])                     # a missing band stays NaN (masked downstream)
    fin = np.isfinite(ref)
    assert fin.sum() == ref.size - 1 and np.abs(got[fin] - ref[fin]).max() < 2e-5
    # contiguous but 16-byte-misaligned views are accepted (copied)
    base = torch.as_tensor(np.concatenate([[0.0], f.reshape(-1)]).astype(np.float32)).cuda()
    mis = base[1:].reshape(f.shape)
    assert mis.data_ptr() % 16 != 0
    got2 = flux_to_abmag(mis, None, 50.0).cpu().double().numpy()
    assert np.array_equal(np.isnan(got2), np.isnan(got)) and np.abs(got2[fin] - got[fin]).max() == 0
    ok = np.isfinite(rerr) & (f > 0)
    assert np.abs(merr.cpu().double().numpy()[ok] - rerr[ok]).max() < 1e-4 * np.abs(rerr[ok]).max()


def test_flux_to_asinh_matches_oracle():
    from oracle import features as OF
    from synference_amd.features import flux_to_asinh
    rng = np.random.default_rng(1)
    f = (rng.normal(0, 1, size=(777, 9)) * 10 ** rng.uniform(-1, 5, size=(777, 9))).astype(np.float32)   # both signs
    e = (0.1 * np.abs(f) + 1).astype(np.float32)
    fb = (10 ** rng.uniform(0, 2, size=9)).astype(np.float32)
    mag, merr = flux_to_asinh(torch.as_tensor(f).cuda(), fb, torch.as_tensor(e).cuda())
    rm, re = OF.flux_to_asinh(f, fb, e)
    assert np.abs(mag.cpu().double().numpy() - rm).max() < 5e-5
    assert np.abs(merr.cpu().double().numpy() - re).max() < 1e-5 * max(1.0, np.abs(re).max())
    m_scalar = flux_to_asinh(torch.as_tensor(f).cuda(), 5.0)
    assert np.abs(m_scalar.cpu().double().numpy() - OF.flux_to_asinh(f, 5.0)).max() < 5e-5


def test_scatter_depths_matches_oracle_draw_for_draw():
    from oracle import features as OF
    from synference_amd.features import scatter_depths
    rng = np.random.default_rng(2)
    f = rng.uniform(-5, 200, size=(501, 10)).astype(np.float32)
    depths = rng.uniform(1, 30, size=10).astype(np.float32)
    out, err = scatter_depths(torch.as_tensor(f).cuda(), depths, n_scatters=3, depth_sigma=5.0, min_flux_pc_error=1.5,
                              seed=77, return_errors=True)
    ro, rs = OF.scatter_depths(f, depths, 3, 5.0, 1.5, 77)
    assert out.shape == (1503, 10)
    assert np.abs(err.cpu().double().numpy() - rs).max() < 1e-5
    assert np.abs(out.cpu().double().numpy() - ro).max() < 2e-4        # same Philox stream, fp32 Box-Muller
    assert scatter_depths(torch.as_tensor(f[:0]).cuda(), depths).shape == (0, 10)


def test_pit_ranks_match_oracle():
    from oracle import features as OF
    from synference_amd.features import pit_ranks
    rng = np.random.default_rng(3)
    s = rng.normal(size=(37, 300, 4)).astype(np.float32)
    s[5, :, 2] = np.nan; s[6, ::3, 1] = np.nan
    t = rng.normal(size=(37, 4)).astype(np.float32)
    got = pit_ranks(torch.as_tensor(s).cuda(), torch.as_tensor(t)).cpu().double().numpy()
    ref = OF.pit_ranks(s, t)
    assert np.isnan(got[5, 2]) and np.isnan(ref[5, 2])
    ok = np.isfinite(ref)
    assert np.abs(got[ok] - ref[ok]).max() < 1e-6


def test_calculate_pit_matches_reference_definition(fitted):
    f = fitted[0]
    X, y = f._X_test[:64], f._y_test[:64]
    s = f.sample_posterior(X, num_samples=200, seed=11)
    pit = f.calculate_PIT(X, y, samples=s)
    ref = np.sort(np.array([np.mean(s[i] < y[i]) for i in range(len(y))]))     # sbi_runner.py:7153-7158
    ref = ref / ref[-1]
    assert pit.shape == (64,) and np.abs(pit - ref).max() < 1e-6
    # evaluate_model: the reference's keys and arithmetic (sbi_runner.py:6596-6639) on the same draws
    m = f.evaluate_model(X_test=X, y_test=y, num_samples=200, samples=s)
    mean_pred, median_pred = s.mean(1), np.median(s, 1)
    ss_res, ss_tot = np.sum((y - mean_pred) ** 2, 0), np.sum((y - np.mean(y)) ** 2, 0)
    want = {"MSE": np.mean((y - mean_pred) ** 2, 0), "RMSE": np.sqrt(np.mean((y - mean_pred) ** 2, 0)),
            "mean_ae": np.mean(np.abs(y - mean_pred), 0), "median_ae": np.median(np.abs(y - median_pred), 0),
            "R_squared": 1 - ss_res / ss_tot, "RMSE_norm": np.sqrt(np.mean((y - mean_pred) ** 2, 0)) / np.std(y),
            "mean_ae_norm": np.mean(np.abs(y - mean_pred), 0) / np.std(y)}
    for k, v in want.items():
        assert np.allclose(m[k], v, rtol=2e-5, atol=1e-6), k
    assert abs(m["log_dpit_max"] + 0.5 * np.log(np.max(np.abs(ref - np.linspace(0, 1, 64))))) < 1e-5
    assert np.isfinite(m["mean_log_prob"]) and "tarp" not in m
    pooled = f.evaluate_model(X_test=X, y_test=y, num_samples=200, samples=s, independent_metrics=False)
    assert isinstance(pooled["MSE"], float) and abs(pooled["MSE"] / np.mean((y - mean_pred) ** 2) - 1.0) < 1e-5
    with pytest.raises(ValueError, match="Samples must have 100 samples"):
        f.evaluate_model(X_test=X, y_test=y, num_samples=100, samples=s)


def test_scatter_depths_with_depth_sets():
    """2-D depths (k sets x C bands): one set per band and scatter copy (sbi_runner.py:636-649)."""
    from oracle import features as OF
    from synference_amd.features import scatter_depths
    rng = np.random.default_rng(5)
    f = rng.uniform(1, 100, size=(200, 6)).astype(np.float32)
    sets = rng.uniform(1, 40, size=(4, 6)).astype(np.float32)
    out, err = scatter_depths(torch.as_tensor(f).cuda(), sets, n_scatters=3, depth_sigma=5.0, seed=9, return_errors=True)
    err = err.cpu().numpy().reshape(200, 3, 6)
    assert np.allclose(err, err[0][None])                             # same choice for every object
    for s in range(3):
        for c in range(6):
            assert np.isclose(sets[:, c] / 5.0, err[0, s, c], rtol=1e-6).any()     # each sigma is one of the k sets
    ro, _ = OF.scatter_depths(f, err[0] * 5.0, 3, 5.0, 0.0, 9)        # oracle with the same per-scatter sigma rows
    assert np.abs(out.cpu().double().numpy() - ro).max() < 2e-4


def test_default_batch_training_is_reproducible_run_to_run():
    """At the reference's batch sizes (<= 512) a seeded run_single_sbi reproduces itself bit for bit: the gradient is
    summed in tile order, the clip norm and Adam in a fixed order, the shuffles come from the seeded generator."""
    from synference_amd import SBI_Fitter
    from synference_amd.synthetic import make_catalogue
    x, theta, names = make_catalogue(1500, 10, 5, seed=2)
    outs = []
    for _ in range(2):
        f = SBI_Fitter("rep", names, [f"F{i}" for i in range(10)], feature_array=x, parameter_array=theta)
        post, stats = f.run_single_sbi(model_type="nsf", hidden_features=32, num_transforms=2, training_batch_size=64,
                                       learning_rate=1e-3, stop_after_epochs=2, max_num_epochs=4, random_seed=5,
                                       save_model=False, verbose=False, plot=False)
        outs.append((post.posteriors[0].posterior_estimator.flat.detach().cpu().clone(), stats[0]["training_loss"]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


def test_batched_posterior_api_of_sbi_023(fitted):
    """sbi >= 0.23 surface the reference probes for (ref: custom_runner.py:441-452, 489-493):
    ``sample_batched((S,), x=X)`` -> (S, N, D) and ``log_prob_batched(theta (S,N,D), x (N,C))`` -> (S, N)."""
    f, post, _, _ = fitted
    X = torch.as_tensor(f._X_test[:7])
    p0 = post.posteriors[0]
    s = p0.sample_batched((33,), x=X, seed=4)
    assert s.shape == (33, 7, 5)
    assert torch.equal(s.permute(1, 0, 2).contiguous(), p0.sample_catalogue(X, 33, seed=4))
    lp = p0.log_prob_batched(s, X, norm_posterior=False)
    assert lp.shape == (33, 7) and torch.isfinite(lp).all()
    ref = p0.log_prob_catalogue(s[5], X, norm_posterior=False)          # draw 5 of every observation
    assert torch.allclose(lp[5], ref, atol=1e-5)
    es = post.sample_batched((16,), x=X, seed=2)                        # the ensemble offers the same surface
    assert es.shape == (16, 7, 5) and torch.isfinite(es).all()


def test_library_file_to_trained_posterior(tmp_path):
    """f1 + f2 + the hot path: a library file in the reference's HDF5 layout (ref: library.py:4074-4153) -> device
    feature transform (nJy -> AB magnitudes) -> training -> sampling, without h5py."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "helpers"))
    from hdf5_fixture import write_library
    from synference_amd import SBI_Fitter
    from synference_amd.synthetic import make_catalogue
    x, theta, names = make_catalogue(3000, 10, 5, seed=9)              # x: AB-magnitude-like features (N, C)
    flux_njy = 10.0 ** ((23.9 - x.astype(np.float64)) / 2.5) * 1000.0  # the fluxes behind those magnitudes
    p = str(tmp_path / "grid.hdf5")
    write_library(p, flux_njy.T, theta.T, [f"JWST/NIRCam.F{i}" for i in range(10)], names, [""] * 5)
    f = SBI_Fitter.init_from_hdf5("lib", p)
    feats, fnames = f.create_feature_array_from_raw_photometry(normed_flux_units="AB", norm_mag_limit=50.0)
    assert feats.shape == (3000, 10) and feats.dtype == np.float32 and np.abs(feats - np.minimum(x, 50.0)).max() < 2e-3
    post, stats = f.run_single_sbi(model_type="maf", hidden_features=32, num_transforms=3, training_batch_size=256,
                                   learning_rate=2e-3, stop_after_epochs=2, max_num_epochs=6, random_seed=1,
                                   save_model=False, verbose=False)
    assert stats[0]["training_loss"][-1] < stats[0]["training_loss"][0]
    s = f.sample_posterior(f._X_test[:6], num_samples=50, seed=2)
    assert s.shape == (6, 50, 5) and np.isfinite(s).all()
    # an observed catalogue in the training units goes through create_features_from_observations (sbi_runner.py:3061-3068):
    # short column names mapped to the filter codes, one row with the missing-data flag
    import pandas as pd
    obs = pd.DataFrame({f"F{i}": f._X_test[:8, i].astype(np.float64) for i in range(10)})
    obs.loc[3, "F4"] = -99.0
    cmap = {f"F{i}": f"JWST/NIRCam.F{i}" for i in range(10)}
    table = f.fit_catalogue(obs, columns_to_feature_names=cmap, flux_units="AB", num_samples=200, seed=5)
    direct = f.fit_catalogue(f._X_test[:8], num_samples=200, seed=5, append_to_input=False)
    qcols = [c for c in table.columns if c.endswith(("_16", "_50", "_84"))]
    assert len(qcols) == 15 and list(table.columns[:10]) == list(obs.columns)
    assert table.loc[3, qcols].isna().all() and table.drop(index=3)[qcols].notna().all().all()
    # rows 0-2 sit at the same catalogue positions in both calls: same slots, same draws, same quantiles
    assert np.allclose(table.loc[:2, qcols].to_numpy(float), direct.loc[:2, qcols].to_numpy(float), rtol=1e-6)
    fa, mask = f.fit_catalogue(obs, columns_to_feature_names=cmap, flux_units="AB", return_feature_array=True)
    assert fa.shape == (7, 10) and mask.tolist() == [False, False, False, True, False, False, False, False]
    tq, full = f.fit_catalogue(obs, columns_to_feature_names=cmap, flux_units="AB", num_samples=64, seed=5,
                               return_full_samples=True, timeout_seconds_per_row=30)
    assert full.shape == (8, 64, 5) and np.isnan(full[3]).all() and np.isfinite(np.delete(full, 3, axis=0)).all()


def _library_fitter(C=6, N=400, D=3, seed=4):
    from synference_amd import SBI_Fitter
    rng = np.random.default_rng(seed)
    grid = (10 ** rng.uniform(0.5, 4.5, size=(C, N))).astype(np.float64)           # nJy
    grid[1, 5] = -3.0; grid[2, 6] = 0.0; grid[0, 7] = np.nan
    grid[:, 9] = 1e-40                                                              # a dropout row: every band at the limit
    params = rng.normal(size=(N, D))
    supp = rng.uniform(1, 2, size=(2, N))
    names = [f"F{i}" for i in range(C)]
    f = SBI_Fitter("lib", [f"p{i}" for i in range(D)], raw_observation_names=names, raw_observation_grid=grid,
                   parameter_array=params, parameter_units=["u"] * D, raw_observation_units="nJy",
                   supplementary_parameters=supp, supplementary_parameter_names=["mass", "sfr"],
                   supplementary_parameter_units=["Msun", "Msun/yr"])
    return f, grid, names, params, supp


@pytest.mark.parametrize("opts", [
    dict(),
    dict(scatter_fluxes=3, depths=np.array([30., 40., 50., 60., 70., 80.]), include_errors_in_feature_array=True, min_flux_pc_error=5.0),
    dict(normalize_method="F2", normalization_unit="AB", photometry_to_remove=["F5"]),
    dict(normalize_method="F0", normalization_unit="log10 nJy", scatter_fluxes=2, depths={f"F{i}": 20.0 + i for i in range(6)},
         include_errors_in_feature_array=True, drop_dropouts=True, drop_dropout_fraction=0.8),
    dict(normed_flux_units="asinh", asinh_softening_parameters="SNR_2", scatter_fluxes=2,
         depths=np.array([30., 40., 50., 60., 70., 80.]), include_errors_in_feature_array=True),
    dict(normed_flux_units="asinh", asinh_softening_parameters={f"F{i}": 5.0 + i for i in range(6)}),
])
def test_feature_array_from_raw_photometry_matches_the_oracle_restatement(opts):
    """ref: sbi_runner.py:1429-2222 (AB branch): scatter by depths, magnitudes + errors, colours relative to a filter, the
    normalisation column, row deletions, and the parameter array that goes with it (update_parameter_array, 476-578)."""
    from oracle import features as OFE
    f, grid, names, params, supp = _library_fitter()
    feat, fnames = f.create_feature_array_from_raw_photometry(seed=11, verbose=False, parameters_to_remove=["p1"],
                                                             parameters_to_add=["sfr"],
                                                             parameter_transformations={"p0": np.tanh}, **opts)
    o = dict(opts)
    dep = o.pop("depths", None)
    if isinstance(dep, dict):
        dep = np.array([dep[n] for n in names if n not in o.get("photometry_to_remove", [])])
    fb = o.get("asinh_softening_parameters")
    if isinstance(fb, str):
        fb = float(fb.split("_")[-1]) * np.asarray(dep) / 5.0
    elif isinstance(fb, dict):
        fb = np.array([fb[n] for n in names])
    ref, rnames, deleted = OFE.feature_array_ab(grid, names, asinh_f_b=fb, normalize_method=o.get("normalize_method"),
                                                normalization_unit=o.get("normalization_unit", "AB"),
                                                scatter_fluxes=o.get("scatter_fluxes", 0), depths=dep,
                                                include_errors=o.get("include_errors_in_feature_array", False),
                                                min_flux_pc_error=o.get("min_flux_pc_error", 0.0),
                                                photometry_to_remove=o.get("photometry_to_remove", ()),
                                                drop_dropouts=o.get("drop_dropouts", False),
                                                drop_dropout_fraction=o.get("drop_dropout_fraction", 1.0), seed=11)
    assert list(fnames) == rnames and feat.dtype == np.float32 and feat.shape == ref.shape
    assert len(deleted) >= 1 and 7 * max(int(opts.get("scatter_fluxes", 0)), 1) in deleted     # the NaN band row is gone
    scale = np.maximum(1.0, np.abs(ref))
    err = np.abs((feat - ref) / scale)
    # (a scattered flux that lands near zero loses float32 digits to cancellation before the logarithm: rare, bounded)
    assert np.quantile(err, 0.999) < 2e-5 and err.max() < 2e-3, (np.quantile(err, 0.999), err.max())
    # the parameter array that goes with it
    n_sc = max(int(o.get("scatter_fluxes", 0)), 1)
    want = np.column_stack((np.delete(params, 1, axis=1), supp[1]))
    want = np.delete(np.repeat(want, n_sc, axis=0), deleted, axis=0)
    want[:, 0] = np.tanh(want[:, 0])
    assert f.fitted_parameter_names == ["tanh_p0", "p2", "sfr"] and f.fitted_parameter_units == ["tanh(u)", "u", "Msun/yr"]
    assert f.fitted_parameter_array.shape == (feat.shape[0], 3) and np.allclose(f.fitted_parameter_array, want)
    nb = len(f.feature_array_flags["raw_observation_names"])
    assert f.feature_units[:nb] == [opts.get("normed_flux_units", "AB")] * nb and len(f.feature_units) == feat.shape[1]


def test_feature_array_argument_errors_follow_the_reference():
    f, grid, names, params, supp = _library_fitter()
    with pytest.raises(ValueError, match="No matching photometry filters"):
        f.create_feature_array_from_raw_photometry(photometry_to_remove=["nope"])
    with pytest.raises(ValueError, match="depths or empirical noise models must be provided"):
        f.create_feature_array_from_raw_photometry(scatter_fluxes=2)
    with pytest.raises(NotImplementedError, match="filter name"):
        f.create_feature_array_from_raw_photometry(normalize_method="mass")
    with pytest.raises(ValueError, match="not found in supplementary parameters"):
        f.create_feature_array_from_raw_photometry(parameters_to_add=["age"])
    with pytest.raises(ValueError, match="HIP path"):
        f.create_feature_array_from_raw_photometry(normed_flux_units="log10 nJy")
    with pytest.raises(AssertionError, match="asinh_softening_parameters must be provided"):
        f.create_feature_array_from_raw_photometry(normed_flux_units="asinh")
    feat, _ = f.create_feature_array_from_raw_photometry(max_rows=50, seed=2, verbose=False)
    assert feat.shape == (50, 6) and f.fitted_parameter_array.shape == (50, 3)
    feat2, names2 = f.create_feature_array(flux_units="AB", verbose=False)      # the simple wrapper (sbi_runner.py:1065-1094)
    assert feat2.shape[1] == 6 and names2 == [f"F{i}" for i in range(6)]

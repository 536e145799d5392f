"""The two BASELINE configs that only exist at size, on the GPU.

configs[4]: ensemble of 5 NSF posteriors, batched ``posterior.sample()`` over a 1e5-source catalogue with the bf16
            hidden-layer MFMA path (ref: custom_runner.py:278-285, sbi_runner.py:6438-6442);
configs[3]: 1M-galaxy 20-filter NSF, data-parallel training (the reference's epoch loop,
            ref: custom_runner.py:553-742) -- here with two ranks sharing the one GPU of the test box, through the real
            kernels (HipTrainOps), as fresh child processes.
"""
import dataclasses
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from cases import make_case
from oracle import flows as OF
from oracle import posterior as OP

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg4_ensemble(n_members=5, seed0=40):
    """Five NSF members of the cfg3/cfg4 shape (D=8, C=20, H=50, T=5, K=8), distinct weights, bf16 hidden layers."""
    from synference_amd.estimator import FlowEstimator
    from synference_amd.posterior import EnsemblePosterior, FlowPosterior
    from synference_amd.priors import CustomIndependentUniform
    members, ospecs, flats = [], [], []
    base = make_case("nsf_cfg3", seed=0, B=64, spread=0.3)
    for e in range(n_members):
        ospec, spec, flat, theta, x = make_case("nsf_cfg3", seed=seed0 + e, B=64, spread=0.3)
        # one standardisation for all members (they were "trained" on the same library)
        spec = dataclasses.replace(spec, hidden_bf16=True, theta_mean=base[1].theta_mean, theta_std=base[1].theta_std,
                                   x_mean=base[1].x_mean, x_std=base[1].x_std)
        ospec = dataclasses.replace(ospec, hidden_bf16=True, theta_mean=base[0].theta_mean, theta_std=base[0].theta_std,
                                    x_mean=base[0].x_mean, x_std=base[0].x_std)
        ospecs.append(ospec)
        flats.append(torch.as_tensor(flat))
        members.append((spec, flat))
    x = base[4]
    free, _ = OP.sample(ospecs[0], flats[0], x[:8], 400, 99, dtype=torch.float32)
    lo = np.quantile(free.reshape(-1, 8), 0.02, axis=0).astype(np.float32) - 0.5
    hi = np.quantile(free.reshape(-1, 8), 0.98, axis=0).astype(np.float32) + 0.5
    prior = CustomIndependentUniform(lo, hi, [f"p{i}" for i in range(8)], device="cuda:0")
    posts = [FlowPosterior(FlowEstimator(s, torch.as_tensor(f), device="cuda:0").to("cuda:0"), prior) for s, f in members]
    weights = np.array([0.35, 0.25, 0.2, 0.12, 0.08][:n_members])
    return EnsemblePosterior(posts, weights=weights / weights.sum()), ospecs, flats, weights / weights.sum(), lo, hi, base


def test_cfg4_five_member_bf16_nsf_ensemble_slice_matches_oracle_draw_for_draw():
    ens, ospecs, flats, w, lo, hi, base = _cfg4_ensemble()
    x = base[4][:6]
    S, seed = 200, 17
    got = ens.sample_catalogue(torch.as_tensor(x), S, seed=seed).cpu().double().numpy()
    assert got.shape == (6, S, 8) and np.isfinite(got).all()
    assert ((got >= lo) & (got <= hi)).all()
    ref = OP.ensemble_sample(ospecs, flats, w, x, S, seed & 0xFFFFFFFF, lo, hi, dtype=torch.float32)
    assert np.isfinite(ref).all()
    # bf16 operands: activations on a rounding boundary may round differently in the fp32 (HIP) and fp64 (oracle)
    # producers, so a small fraction of draws differs at the bf16 level; the rest is tight
    err = np.abs((got - ref) / (hi - lo).astype(np.float64)).max(-1)
    assert np.median(err) < 5e-4 and (err > 2e-2).mean() < 0.05, (np.median(err), (err > 2e-2).mean())
    # member e owns positions [cum_{e-1}, cum_e) of every row: counts follow the multinomial split of the weights
    counts = OP.ensemble_counts(w, S, len(x), seed & 0xFFFFFFFF)
    assert counts.shape == (6, 5) and (counts.sum(1) == S).all()
    cum = np.concatenate([np.zeros((6, 1), np.int64), np.cumsum(counts, 1)], 1)
    for e in range(5):                                          # each member alone reproduces its own positions
        single, _ = OP.sample_slots(ospecs[e], flats[e], x, np.concatenate(
            [g * S + np.arange(cum[g, e], cum[g, e + 1]) for g in range(6)]).astype(np.uint64), S, seed & 0xFFFFFFFF, lo, hi)
        own = np.concatenate([got[g, cum[g, e]:cum[g, e + 1]] for g in range(6)])
        e2 = np.abs((own - single) / (hi - lo).astype(np.float64)).max(-1)
        assert np.median(e2) < 1e-3 and (e2 > 2e-2).mean() < 0.25, (e, np.median(e2), (e2 > 2e-2).mean())
    # log_prob mixes the members: logsumexp_i(log w_i + lp_i)
    th = got[:, :3].reshape(-1, 8).astype(np.float32)
    xx = np.repeat(x, 3, 0)
    lp = ens.log_prob_catalogue(torch.as_tensor(th), torch.as_tensor(xx), norm_posterior=False).cpu().double().numpy()
    rlp = OP.ensemble_log_prob(ospecs, flats, w, th, xx, lo, hi)
    assert np.abs(lp - rlp).max() < 0.25 and np.median(np.abs(lp - rlp)) < 5e-3     # documented bf16 tolerance


def test_cfg4_full_size_catalogue_1e5_sources_x_1000_draws():
    """BASELINE configs[4] at size on one GPU: 1e5 sources x 1000 draws, 5 bf16 NSF members; only device-side
    summaries cross PCIe.  Size-independent properties + bit-equality of a slice with a small call."""
    from synference_amd.posterior import device_quantiles
    from synference_amd.synthetic import make_catalogue
    ens, ospecs, flats, w, lo, hi, base = _cfg4_ensemble()
    N, S, seed = 100_000, 1000, 5
    rng = np.random.default_rng(8)
    X = (rng.normal(size=(N, 20)) * base[1].x_std + base[1].x_mean).astype(np.float32)
    Xd = torch.as_tensor(X).cuda()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    s = ens.sample_catalogue(Xd, S, seed=seed)                # (N, S, 8) float32 on the device: 3.2 GB
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert s.shape == (N, S, 8) and s.dtype == torch.float32
    assert bool(torch.isfinite(s).all())
    lo_t, hi_t = torch.as_tensor(lo).cuda(), torch.as_tensor(hi).cuda()
    assert bool(((s >= lo_t) & (s <= hi_t)).all())
    q = device_quantiles(s, (0.16, 0.5, 0.84))               # (N, 8, 3)
    assert q.shape == (N, 8, 3) and bool(torch.isfinite(q).all())
    assert bool((q[..., 0] <= q[..., 1]).all()) and bool((q[..., 1] <= q[..., 2]).all())
    # the first rows of the big call are, bit for bit, what a small call over just those rows returns (slots, member
    # split and Philox streams do not depend on the catalogue size) -- and the small call is oracle-checked above
    small = ens.sample_catalogue(Xd[:48].clone(), S, seed=seed)
    assert torch.equal(small, s[:48])
    # each member's share of the draws follows its weight
    counts = OP.ensemble_counts(w, S, 2000, seed & 0xFFFFFFFF)
    assert np.abs(counts.mean(0) / S - w).max() < 0.01
    print(f"cfg4 full size: {N * S / dt / 1e6:.1f} M accepted draws/s (5-member bf16 NSF ensemble, 1 GPU)")


def test_cfg3_data_parallel_training_on_the_hip_kernels(tmp_path):
    """Two fresh child ranks (gloo, both on cuda:0) run train_flow with HipTrainOps on a 1e6-row NSF mock.  The ranks read
    the catalogue from a library FILE in the reference's on-disk format (SURVEY 8f f1 at size: 1e6 x (20 + 8) float64,
    chunked + deflate + shuffle, written here by the test-side writer) through synference_amd.hdf5_lite."""
    from synference_amd.engine import HipFlow
    from synference_amd.estimator import build_flow
    from synference_amd.synthetic import make_catalogue
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers.hdf5_fixture import write_library
    x_all, theta_all, names_all = make_catalogue(1_000_000, 20, 8, seed=11)
    h5 = str(tmp_path / "cfg3_library.h5")
    write_library(h5, np.ascontiguousarray(x_all.T).astype(np.float64), np.ascontiguousarray(theta_all.T).astype(np.float64),
                  [f"F{i}" for i in range(20)], list(names_all), chunks=(4, 8192), gzip=1, shuffle=True)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SF_DP_OUT=str(tmp_path), SF_DP_H5=h5, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "dp_child.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    # ranks end bit-identical although rank 1 started elsewhere with another seed
    assert torch.equal(r0["flat"], r1["flat"])
    s0, s1 = r0["summary"], r1["summary"]
    assert s0["training_loss"] == s1["training_loss"] and s0["validation_loss"] == s1["validation_loss"]
    assert np.isfinite(s0["training_loss"]).all() and s0["training_loss"][-1] < s0["training_loss"][0]
    assert (r0["flat"] - r0["flat0"]).abs().max() > 1e-3               # it trained
    # the all-reduced sharded gradient == the single-process gradient of the global batch
    assert torch.equal(r0["grad"], r1["grad"])
    print([ln for o in outs for ln in o.splitlines() if ln.startswith("library file")][:1])
    x, theta = x_all, theta_all
    dev = torch.device("cuda:0")
    est = build_flow("nsf", theta[:20000], x[:20000], hidden_features=50, num_transforms=5, num_bins=8, device=dev,
                     generator=torch.Generator().manual_seed(3)).to(dev)
    assert torch.equal(est.flat.detach().cpu(), r0["flat0"])
    rows = r0["rows"].to(dev)
    g = torch.empty_like(est.flat.data)
    est.flow.loss_grad_rows(est.flat.data, torch.as_tensor(theta, dtype=torch.float32).to(dev), torch.as_tensor(x).to(dev),
                            rows, 1.0 / rows.numel(), g)
    ref = g.cpu()
    assert (r0["grad"] - ref).abs().max() < 1e-5 * max(1.0, ref.abs().max().item())


def test_rank_sharded_catalogue_evaluation_equals_the_single_process_call(tmp_path):
    """catalogue rows are sharded over the ranks as contiguous blocks (no data-path collective) and gathered.
    Two fresh child ranks (gloo, both on cuda:0) run SBI_Fitter.sample_posterior / log_prob / fit_catalogue under a
    process group; every rank must return, bit for bit, what this single process returns."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
    import shard_child
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SF_DP_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "shard_child.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r0 = torch.load(tmp_path / "shard_rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "shard_rank1.pt", weights_only=False)
    for kind in ("maf", "nsf"):
        f, x, theta = shard_child.build(kind)
        ref = shard_child.run(f, x, theta)
        assert np.isfinite(ref["samples"]).all() and np.isfinite(ref["lp"]).any()   # (-inf: theta outside the box)
        for k in ("samples", "lp", "table"):
            assert np.array_equal(r0[kind][k], ref[k], equal_nan=True), (kind, k, "rank 0 vs single process")
        # rank 1 keeps its own block of the draws by default (one copy of the big array, on rank 0) ...
        a1, b1 = r1[kind]["rows"]
        assert (a1, b1) == (50, 101) and np.array_equal(r1[kind]["samples"], ref["samples"][a1:b1], equal_nan=True)
        # ... and the small outputs, and gather="all", are the whole thing on every rank
        for k in ("lp", "table", "table_rs", "samples_rs", "table_hq"):
            assert np.array_equal(r1[kind][k], ref[k], equal_nan=True), (kind, k, "rank 1 vs single process")
            assert np.array_equal(r0[kind][k], ref[k], equal_nan=True), (kind, k, "rank 0 vs single process")
        for r in (r0, r1):
            assert np.array_equal(r[kind]["samples_all"], ref["samples"], equal_nan=True)
    # and the draws of a block do not depend on how the catalogue is cut: rows 40..60 alone, keyed by their position
    f, x, theta = shard_child.build("maf")
    whole = f.posteriors.sample_catalogue(torch.as_tensor(x[2000:2101]), 64, 17)
    part = f.posteriors.sample_catalogue(torch.as_tensor(x[2040:2060]), 64, 17, row_offset=40)
    assert torch.equal(whole[40:60], part)


@pytest.mark.parametrize("name", ["maf_cfg1", "nsf_cfg3", "maf_wide"])
def test_large_batch_gradient_path_matches_autograd(name):
    """Batches above 512 rows take the many-tiles accumulation path of the training kernels (gradient-image replicas
    reduced across workgroups); it must agree with fp64 autograd on the oracle like the small-batch path does."""
    from synference_amd.engine import HipFlow
    from test_gpu_train import oracle_loss_grad
    B = 2048 + 37
    ospec, spec, flat, theta, x = make_case(name, B=B)
    f = HipFlow(spec, "cuda:0")
    loss, grad = f.loss_grad(torch.as_tensor(flat), theta, x, 1.0 / B)
    rloss, rgrad = oracle_loss_grad(ospec, flat, theta, x)
    assert np.abs(loss.cpu().double().numpy() - rloss).max() < 1e-4
    denom = np.abs(rgrad).max()
    g = grad.cpu().double().numpy()
    assert np.abs(g - rgrad).max() < 2e-4 * denom
    for n, s, o in OF.param_layout(ospec):
        k = int(np.prod(s))
        assert np.abs(g[o:o + k] - rgrad[o:o + k]).max() < 2e-4 * denom + 1e-7, n
    # and a second call gives the same answer to accumulation-order noise
    _, grad2 = f.loss_grad(torch.as_tensor(flat), theta, x, 1.0 / B)
    assert (grad2 - grad).abs().max().item() < 1e-5 * max(1.0, float(denom))

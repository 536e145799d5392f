"""Host logic on CPU: priors (ref: custom_runner.py:971-1207), the epoch loop's bookkeeping and the
data-parallel path under gloo (world_size 2) with a test double in place of the HIP kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import flows as OF
from synference_amd.priors import CustomIndependentUniform, prior_from_parameters
from synference_amd.runner import finish_summary, split_indices, train_flow
from synference_amd.spec import FlowSpec, init_params, num_params, param_layout, zscore_stats


def test_prior_box_predicate_and_log_prob():
    p = CustomIndependentUniform([0.0, -1.0], [2.0, 1.0], ["a", "b"])
    v = torch.tensor([[0.0, -1.0], [2.0, 1.0], [1.0, 1.0001], [-1e-6, 0.0]])
    assert p.support.check(v).tolist() == [True, True, False, False]   # closed box (custom_runner.py:986)
    lp = p.log_prob(v)
    assert lp[0].item() == pytest.approx(-np.log(4.0))
    assert np.isneginf(lp[1].item())                                   # high is exclusive in log_prob (1107)
    s = p.sample((1000,))
    assert s.shape == (1000, 2) and p.support.check(s).all()
    with pytest.raises(ValueError):
        CustomIndependentUniform([0.0], [1.0], ["a", "b"])
    assert "a" in p.acceptance_report(v)


def test_prior_from_parameters_follows_create_priors():
    th = np.array([[1.0, 10.0], [3.0, 30.0], [2.0, 20.0]])
    p = prior_from_parameters(th, ["m", "z"])
    assert p.low.tolist() == [1.0, 10.0] and p.high.tolist() == [3.0, 30.0]
    p = prior_from_parameters(th, ["m", "z"], override={"z": (0.0, 50.0)}, extend_pc=10.0)
    assert p.low.tolist() == pytest.approx([0.8, 0.0]) and p.high.tolist() == pytest.approx([3.2, 50.0])
    with pytest.raises(ValueError, match="zero"):
        prior_from_parameters(np.ones((3, 1)), ["c"])


def test_spec_layout_matches_oracle_layout_and_counts():
    for kind, D, C, K in (("maf", 5, 10, 10), ("nsf", 8, 20, 8), ("nsf", 5, 3, 10)):
        s = FlowSpec(kind=kind, D=D, C=C, K=K)
        o = OF.FlowSpec(kind=kind, D=D, C=C, K=K)
        assert param_layout(s) == OF.param_layout(o)
        flat = init_params(s, torch.Generator().manual_seed(0))
        assert flat.numel() == num_params(s) == OF.num_params(o)
    st = zscore_stats(np.random.default_rng(0).normal(size=(50, 3)), np.ones((50, 2)))
    assert st["x_std"].tolist() == pytest.approx([1e-7, 1e-7])


def test_split_and_summary_conventions():
    g = torch.Generator().manual_seed(0)
    tr, va = split_indices(103, 0.1, g)
    assert len(va) == 10 and len(tr) == 93 and len(set(tr.tolist()) | set(va.tolist())) == 103
    s = finish_summary({"training_loss": [2.0, 1.0], "validation_loss": [2.5, 1.5], "best_validation_loss": [1.5]})
    assert s["training_log_probs"] == [-2.0, -1.0] and s["best_validation_log_prob"] == [-1.5]


# ------------------------------------------------------------------------------------------------
# test double for the HIP kernels: oracle autograd + torch clip/Adam (tests may use the oracle)
# ------------------------------------------------------------------------------------------------
class OracleOps:
    def __init__(self, ospec):
        self.ospec = ospec
        self.flat = None

    def loss_grad(self, flat, theta, x, scale, grad_out):
        p = flat.detach().double().requires_grad_(True)
        loss = -OF.log_prob(self.ospec, p, theta.double(), x.double())
        (loss.sum() * scale).backward()
        grad_out.copy_(p.grad.float())
        return loss.detach().float()

    def refresh(self, flat):
        self.flat = flat.detach().clone()

    def log_prob(self, theta, x):
        with torch.no_grad():
            return OF.log_prob(self.ospec, self.flat.double(), theta.double(), x.double()).float()

    def make_optimizer(self, flat, lr, weight_decay, decoupled):
        class _Opt:
            def __init__(s):
                s.p = torch.nn.Parameter(flat)   # shares storage with the estimator's parameter
                s.o = (torch.optim.AdamW if decoupled else torch.optim.Adam)([s.p], lr=lr, weight_decay=weight_decay)

            def step(s, grad, max_norm):
                s.p.grad = grad.clone()
                if max_norm:
                    torch.nn.utils.clip_grad_norm_([s.p], max_norm)
                s.o.step()

            def state_dict(s):
                return {"exp_avg": torch.zeros(1), "exp_avg_sq": torch.zeros(1), "step": 0}

            def load_state_dict(s, sd):
                pass
        return _Opt()


class _Est(torch.nn.Module):
    def __init__(self, spec):
        super().__init__()
        self.spec = spec
        self.flat = torch.nn.Parameter(init_params(spec, torch.Generator().manual_seed(1)))
        self._packed_version = None


def _toy(n=600, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, 3)).astype(np.float32)
    th = np.stack([x[:, 0] + 0.1 * rng.normal(size=n), x[:, 1] * x[:, 2] + 0.1 * rng.normal(size=n)], 1).astype(np.float32)
    return torch.as_tensor(th), torch.as_tensor(x)


def _specs(th, x):
    st = zscore_stats(th, x)
    perms = np.array([[1, 0], [0, 1]])
    s = FlowSpec(kind="maf", D=2, C=3, H=8, T=2, perms=perms, **st)
    o = OF.FlowSpec(kind="maf", D=2, C=3, H=8, T=2, perms=perms, **{k: v.astype(np.float64) for k, v in st.items()})
    return s, o


def test_epoch_loop_bookkeeping_single_process(tmp_path):
    th, x = _toy()
    s, o = _specs(th, x)
    est = _Est(s)
    out = train_flow(est, th, x, batch_size=64, learning_rate=5e-3, validation_fraction=0.1, stop_after_epochs=2,
                     max_num_epochs=5, seed=0, ops=OracleOps(o), log_every=0, save_dir=str(tmp_path) + "/m_")
    n_ep = out["epochs_trained"][0]
    assert len(out["training_loss"]) == len(out["validation_loss"]) == n_ep and 2 <= n_ep <= 6
    assert out["training_loss"][-1] < out["training_loss"][0]
    assert out["best_validation_loss"][0] == min(out["validation_loss"])
    assert not os.path.exists(str(tmp_path) + "/m_checkpoint_posterior.pt")   # removed after success
    # epoch average = sum of per-sample losses / (num_batches * batch_size), drop_last (540 rows -> 8 batches)
    assert out["pairs_per_sec"] > 0


def _dp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    th, x = _toy(400, seed=3)
    s, o = _specs(th, x)
    est = _Est(s)
    if rank == 1:   # different start: the runner must broadcast rank 0's parameters
        with torch.no_grad():
            est.flat.add_(1.0)
    out = train_flow(est, th, x, batch_size=32, learning_rate=5e-3, validation_fraction=0.1, stop_after_epochs=50,
                     max_num_epochs=2, seed=5, ops=OracleOps(o), log_every=0)
    ret[rank] = (est.flat.detach().clone(), out)
    dist.destroy_process_group()


def test_data_parallel_gloo_world2_matches_single_process_global_batch():
    """Two ranks with batch 32 each == one process with batch 64 over the same index order is not
    required (shards differ); what must hold: both ranks end bit-identical, losses are finite and the
    epoch averages agree across ranks (they are all-reduced)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    (f0, o0), (f1, o1) = ret[0], ret[1]
    assert torch.equal(f0, f1), "ranks diverged: gradient all-reduce / identical update broken"
    assert o0["training_loss"] == o1["training_loss"] and o0["validation_loss"] == o1["validation_loss"]
    assert np.isfinite(o0["training_loss"]).all() and o0["training_loss"][-1] < o0["training_loss"][0]


def _dp_equiv_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    th, x = _toy(128, seed=4)
    s, o = _specs(th, x)
    ops = OracleOps(o)
    flat = init_params(s, torch.Generator().manual_seed(2))
    g = torch.empty_like(flat)
    half = slice(rank * 64, (rank + 1) * 64)
    ops.loss_grad(flat, th[half], x[half], 1.0 / 128, g)
    dist.all_reduce(g)
    ret[rank] = g.clone()
    dist.destroy_process_group()


def test_sharded_gradient_sum_equals_full_batch_gradient():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_equiv_worker, args=(2, 31500 + (os.getpid() % 2000), ret), nprocs=2, join=True)
    th, x = _toy(128, seed=4)
    s, o = _specs(th, x)
    flat = init_params(s, torch.Generator().manual_seed(2))
    g = torch.empty_like(flat)
    OracleOps(o).loss_grad(flat, th, x, 1.0 / 128, g)
    assert (ret[0] - g).abs().max() < 1e-6 and torch.equal(ret[0], ret[1])


# ------------------------------------------------------------------------------------------------
# round 2: seed / parameter / resume agreement across ranks, rejection schedules of the oracle
# ------------------------------------------------------------------------------------------------
def _dp_seed_worker(rank, world, port, ret, ckpt_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    th, x = _toy(400, seed=3)
    s, o = _specs(th, x)
    est = _Est(s)
    seen = []

    class SpyOps(OracleOps):
        def loss_grad(self, flat, theta, x, scale, grad_out):
            seen.append(theta.clone())
            return super().loss_grad(flat, theta, x, scale, grad_out)

    # seed=None on rank 1 (a per-rank time() seed in round 1), a fixed seed on rank 0: rank 0's must win.
    # Only rank 0 can see the checkpoint directory: the resume decision must be rank 0's for everybody.
    out = train_flow(est, th, x, batch_size=32, learning_rate=5e-3, validation_fraction=0.25, stop_after_epochs=50,
                     max_num_epochs=3, seed=(77 if rank == 0 else None), ops=SpyOps(o), log_every=0,
                     save_dir=(ckpt_dir if rank == 0 else ckpt_dir + "_not_there/"))
    rows = torch.cat(seen)
    ret[rank] = (rows, out["epochs_trained"][0], est.flat.detach().clone())
    dist.destroy_process_group()


def test_data_parallel_ranks_share_split_seed_and_resume_decision(tmp_path):
    th, x = _toy(400, seed=3)
    s, o = _specs(th, x)
    # a checkpoint only rank 0 can see: epoch 3 of a previous run
    ck = str(tmp_path) + "/m_"
    flat = init_params(s, torch.Generator().manual_seed(9))
    torch.save({"epoch": 3, "model_state_dict": {"flat": flat}, "optimizer_state_dict": {"exp_avg": torch.zeros(1),
                "exp_avg_sq": torch.zeros(1), "step": 0}, "train_loss": [3.0, 2.0, 1.5], "val_loss": [3.0, 2.0, 1.6],
                "epochs_since_improvement": 0, "best_val_loss": 1.6, "best_model_state_dict": {"flat": flat}},
               ck + "checkpoint_posterior.pt")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_seed_worker, args=(2, 33500 + (os.getpid() % 2000), ret, ck), nprocs=2, join=True)
    (rows0, ep0, f0), (rows1, ep1, f1) = ret[0], ret[1]
    assert ep0 == ep1 == 4                      # both resumed at epoch 3 and ran one more
    assert torch.equal(f0, f1)
    # same split on both ranks (rank 0's seed): the training rows the two ranks saw are disjoint, and none of them is
    # a validation row of that split
    g = torch.Generator().manual_seed(77)
    tr, va = split_indices(400, 0.25, g)
    as_set = lambda t: {tuple(np.round(r, 6)) for r in t.numpy().tolist()}
    s0, s1, sva = as_set(rows0), as_set(rows1), as_set(th[va])
    assert not (s0 & s1) and not (s0 & sva) and not (s1 & sva)
    assert len(s0) + len(s1) <= len(tr)


def test_oracle_batch_accept_reject_sampler_and_uncapped_slot_schedule():
    from oracle import posterior as OP
    th, x = _toy(50, seed=1)
    s, o = _specs(th, x)
    flat = init_params(s, torch.Generator().manual_seed(4))
    free, _ = OP.sample(o, flat, x[:3].numpy(), 500, 3)
    lo = np.quantile(free.reshape(-1, 2), 0.3, axis=0).astype(np.float32)
    hi = np.quantile(free.reshape(-1, 2), 0.7, axis=0).astype(np.float32)
    # sbi-shaped batch loop: exactly S rows, all inside the box, acceptance consistent with the box's mass
    gen = torch.Generator().manual_seed(0)
    smp, rate, warned = OP.accept_reject_sample(o, flat, x[0].numpy(), 700, lo, hi, gen)
    assert smp.shape == (700, 2) and ((smp >= lo) & (smp <= hi)).all() and 0.05 < rate < 0.5 and not warned
    # per-slot schedule without a ceiling: everything is filled, attempts follow a geometric law with that rate
    out, used = OP.sample_slots(o, flat, x[:1].numpy(), np.arange(400, dtype=np.uint64), 400, 7, lo, hi)
    assert np.isfinite(out).all() and used.max() > 8
    assert abs(400.0 / used.sum() - rate) < 0.05
    # both samplers target the same distribution
    from scipy import stats
    assert min(stats.ks_2samp(out[:, d], smp[:, d]).pvalue for d in range(2)) > 1e-3
    # the progress rule: a dead galaxy (NaN context) ends as NaN rows after the window [0, 1024); a ceiling is a ceiling
    xx = x[:2].numpy().copy()
    xx[1] = np.nan
    out, used = OP.sample_slots(o, flat, xx, np.arange(240, dtype=np.uint64), 120, 7, lo, hi)
    assert np.isfinite(out[:120]).all() and np.isnan(out[120:]).all() and (used[120:] == 1024).all()
    out, used = OP.sample_slots(o, flat, xx, np.arange(20, dtype=np.uint64), 10, 7, lo, hi, max_attempts=5)
    assert np.isnan(out[10:]).all() and used.max() <= 5


def test_bench_starts_its_own_ranks_when_asked_for_several_gpus(monkeypatch):
    """bench.py --gpus N without a launcher environment must start N ranks through torch.distributed.run (and never
    report n_gpus = 1 for it, as round 1 did)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(bench)
    calls = {}

    def fake_run(cmd, env=None, **kw):
        calls["cmd"], calls["env"] = cmd, env

        class R:
            returncode = 0
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = calls["cmd"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "--master-addr" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher whose world size disagrees with --gpus the script refuses instead of mis-reporting
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


def test_custom_config_yaml_maps_fixed_params(tmp_path, monkeypatch):
    """ref: sbi_runner.py:4570-4597 + custom_runner.py:298-365: train_args.fixed_params drive one training run."""
    from synference_amd import SBI_Fitter
    import synference_amd.fitter as fitter_mod
    cfg = tmp_path / "best.yaml"
    cfg.write_text("train_args:\n  skip_optimization: True\n  validation_fraction: 0.1\n  fixed_params:\n"
                   "    model_choice: \"nsf\"\n    optimizer_choice: \"AdamW\"\n    learning_rate: 0.0003\n"
                   "    training_batch_size: 52\n    stop_after_epochs: 47\n    clip_max_norm: 4.7\n"
                   "    nsf_hidden_features: 69\n    nsf_num_transforms: 15\n    nsf_num_bins: 8\n")
    seen = {}

    class FakeRunner:
        @classmethod
        def load(cls, **kw):
            seen.update(kw)
            return cls()

        def __call__(self, loader, seed=None):
            return "posterior", [{"ok": True}]

    monkeypatch.setattr(fitter_mod, "HIPRunner", FakeRunner)
    rng = np.random.default_rng(0)
    f = SBI_Fitter("y", ["a", "b"], ["F0", "F1", "F2"], feature_array=rng.normal(size=(50, 3)),
                   parameter_array=rng.normal(size=(50, 2)))
    post, stats = f.run_single_sbi(custom_config_yaml=str(cfg), verbose=False, save_model=False, random_seed=1, evaluate_model=False)
    ta = seen["train_args"]
    assert ta["training_batch_size"] == 52 and ta["stop_after_epochs"] == 47 and ta["optimizer_choice"] == "AdamW"
    assert abs(ta["learning_rate"] - 3e-4) < 1e-12 and abs(ta["clip_max_norm"] - 4.7) < 1e-12 and ta["validation_fraction"] == 0.1
    net = seen["nets"][0]
    assert net.model == "nsf" and net.model_args == dict(hidden_features=69, num_transforms=15, num_bins=8)
    bad = tmp_path / "search.yaml"
    bad.write_text("train_args:\n  optuna: {n_trials: 3}\n")
    with pytest.raises(ValueError, match="Optuna"):
        f.run_single_sbi(custom_config_yaml=str(bad), verbose=False)


# the reference's run_single_sbi parameters, in order, with their defaults (sbi_runner.py:4392-4435; callables / objects by name)
REF_RUN_SINGLE_SBI = [
    ("train_test_fraction", 0.8), ("random_seed", None), ("backend", "sbi"), ("engine", "NPE"), ("train_indices", None),
    ("test_indices", None), ("n_nets", 1), ("model_type", "mdn"), ("hidden_features", 50), ("num_components", 4),
    ("num_transforms", 4), ("training_batch_size", 64), ("learning_rate", 1e-4), ("validation_fraction", 0.2),
    ("stop_after_epochs", 15), ("clip_max_norm", 5.0), ("additional_model_args", {}), ("save_model", True), ("verbose", True),
    ("prior_method", "ili"), ("out_dir", "<code_path>/models/"), ("plot", True), ("name_append", "timestamp"),
    ("feature_scalar", "StandardScaler"), ("target_scalar", "StandardScaler"), ("set_self", True), ("learning_type", "offline"),
    ("simulator", None), ("num_simulations", 1000), ("num_online_rounds", 5), ("initial_training_from_library", False),
    ("override_prior_ranges", {}), ("online_training_xobs", None), ("load_existing_model", True), ("use_existing_indices", True),
    ("evaluate_model", True), ("save_method", "joblib"), ("num_posterior_draws_per_sample", 1000), ("embedding_net", "Identity"),
    ("custom_config_yaml", None), ("sql_db_path", None)]


def test_run_single_sbi_signature_is_the_references():
    """Positional order, names and defaults of SBI_Fitter.run_single_sbi."""
    import inspect
    from synference_amd import SBI_Fitter
    sig = inspect.signature(SBI_Fitter.run_single_sbi)
    pos = [p for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD and p.name != "self"]
    assert [p.name for p in pos] == [n for n, _ in REF_RUN_SINGLE_SBI]
    for p, (name, default) in zip(pos, REF_RUN_SINGLE_SBI):
        if name == "out_dir":
            assert p.default.endswith("/models/")
        elif name in ("feature_scalar", "target_scalar"):
            assert p.default.__name__ == "StandardScaler"
        elif name == "embedding_net":
            assert p.default is None or type(p.default).__name__ == "Identity"   # (None = identity here: no module built at import)
        else:
            assert p.default == default, (name, p.default, default)
    extra = [p.name for p in sig.parameters.values() if p.kind == p.KEYWORD_ONLY]
    assert extra == ["max_num_epochs", "optimizer_choice"]


def test_run_single_sbi_existing_model_indices_and_manual_prior(tmp_path, monkeypatch):
    """load_existing_model / use_existing_indices / prior_method='manual' (sbi_runner.py:4543-4563, 4617-4636, 4664-4690),
    with a stand-in for the trainer (the host logic only)."""
    from synference_amd import SBI_Fitter
    import synference_amd.fitter as fitter_mod
    seen = {"calls": 0}

    class FakeRunner:
        @classmethod
        def load(cls, **kw):
            seen.update(kw)
            return cls()

        def __call__(self, loader, seed=None):
            seen["calls"] += 1
            seen["x"], seen["theta"] = np.asarray(loader.get_all_data()), np.asarray(loader.get_all_parameters())
            return "posterior", [{"ok": True}]

    monkeypatch.setattr(fitter_mod, "HIPRunner", FakeRunner)
    rng = np.random.default_rng(0)
    x, th = rng.normal(2.0, 3.0, size=(60, 3)), rng.normal(-1.0, 0.5, size=(60, 2))
    f = SBI_Fitter("m", ["a", "b"], ["F0", "F1", "F2"], feature_array=x, parameter_array=th)
    f.run_single_sbi(model_type="maf", save_model=False, evaluate_model=False, verbose=False, random_seed=4, plot=False)
    tr0 = np.array(f._train_indices)
    f.run_single_sbi(model_type="maf", save_model=False, evaluate_model=False, verbose=False, random_seed=5, plot=False)
    assert np.array_equal(tr0, f._train_indices)                      # the stored split is re-used ...
    f.run_single_sbi(model_type="maf", save_model=False, evaluate_model=False, verbose=False, random_seed=5, plot=False,
                     use_existing_indices=False)
    assert not np.array_equal(tr0, f._train_indices)                  # ... unless the caller says otherwise
    # manual prior: the trainer sees standardised arrays, the box is min / max -+ 3 sigma of the scaled parameters
    f.run_single_sbi(model_type="maf", save_model=False, evaluate_model=False, verbose=False, plot=False, prior_method="manual")
    assert np.abs(seen["x"].mean(0)).max() < 1e-5 and np.abs(seen["x"].std(0) - 1).max() < 1e-5
    assert np.abs(seen["theta"].mean(0)).max() < 1e-6
    ys = seen["theta"]
    assert np.allclose(seen["prior"].low.cpu().numpy(), ys.min(0) - 3 * ys.std(0), atol=1e-5)
    assert np.allclose(seen["prior"].high.cpu().numpy(), ys.max(0) + 3 * ys.std(0), atol=1e-5)
    back = f._target_scalar.inverse_transform(ys)
    assert np.allclose(back, th[f._train_indices], atol=1e-9)
    with pytest.raises(ValueError, match="Invalid prior method"):
        f.run_single_sbi(model_type="maf", prior_method="other", save_model=False)
    # an existing {out_dir}/{name}/{name}_{append}_params.pkl: refused (None) or loaded, never silently retrained
    d = tmp_path / "m"
    d.mkdir()
    (d / "m_v1_params.pkl").write_bytes(b"")
    n = seen["calls"]
    assert f.run_single_sbi(model_type="maf", out_dir=str(tmp_path), name_append="v1", load_existing_model=False, plot=False) is None
    monkeypatch.setattr(SBI_Fitter, "load_model_from_pkl", lambda self, path, set_self=True: ("loaded:" + path, ["s"], {}))
    post, stats = f.run_single_sbi(model_type="maf", out_dir=str(tmp_path), name_append="v1", plot=False)
    assert post == "loaded:" + str(d / "m_v1_posterior.pkl") and seen["calls"] == n


@pytest.mark.parametrize("kind,D,C,K,NB", [("maf", 5, 10, 10, 2), ("nsf", 8, 20, 8, 2), ("nsf", 5, 3, 10, 1), ("maf", 3, 4, 10, 3)])
def test_upstream_state_dict_importer_round_trip(kind, D, C, K, NB):
    """synference_amd.importer maps nflows module paths onto the flat vector: a state dict laid out
    with those paths -- under an arbitrary wrapper prefix, with the mask buffers nflows also stores -- comes back as the
    same FlowSpec and flat vector, and the oracle evaluates both identically."""
    from synference_amd.importer import spec_and_flat_from_state_dict, state_dict_from_flat
    rng = np.random.default_rng(1)
    g = torch.Generator().manual_seed(3)
    perms = np.stack([rng.permutation(D) for _ in range(3)]) if kind == "maf" else None
    s = FlowSpec(kind=kind, D=D, C=C, H=24, T=3, K=K, NB=NB, perms=perms, theta_mean=rng.normal(size=D),
                 theta_std=rng.uniform(0.5, 2, size=D), x_mean=rng.normal(size=C), x_std=rng.uniform(0.5, 2, size=C))
    flat = init_params(s, g).numpy() + 0.05 * rng.normal(size=num_params(s)).astype(np.float32)
    sd = state_dict_from_flat(s, flat, prefix="posterior_estimator.net.")   # (with the connectivity buffers nflows stores)
    sd["posterior_estimator.net._distribution._log_z"] = np.zeros(1)
    s2, flat2 = spec_and_flat_from_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    assert (s2.kind, s2.D, s2.C, s2.H, s2.T, s2.NB) == (kind, D, C, 24, 3, NB) and (kind == "maf" or s2.K == K)
    assert np.array_equal(flat2, flat.astype(np.float32))
    assert np.allclose(s2.theta_mean, s.theta_mean, rtol=1e-6) and np.allclose(s2.theta_std, s.theta_std, rtol=1e-6)
    assert np.allclose(s2.x_mean, s.x_mean) and np.allclose(s2.x_std, s.x_std)
    if kind == "maf":
        assert np.array_equal(s2.perms, s.perms)
    with pytest.raises(KeyError, match="not found|expects"):
        bad = {k: v for k, v in sd.items() if not k.endswith("final_layer.bias")}
        spec_and_flat_from_state_dict(bad)
    # upstream's own connectivity buffers are compared with the engine's: a checkpoint wired differently is refused
    from oracle import flows as OF
    if kind == "maf":
        from synference_amd.importer import made_masks
        for ours, ref in zip(made_masks(D, 24), OF.made_masks(D, 24)):
            assert np.array_equal(ours, np.asarray(ref) != 0)           # importer's rule == the oracle's masks
        key = "posterior_estimator.net._transform._transforms.1._transforms.2.autoregressive_net.blocks.0.linear.mask"
        tampered = dict(sd)
        tampered[key] = np.tril(np.ones((24, 24), np.float32))
        with pytest.raises(ValueError, match="MADE mask"):
            spec_and_flat_from_state_dict(tampered)
    else:
        key = "posterior_estimator.net._transform._transforms.1._transforms.0.transform_features"
        tampered = dict(sd)
        tampered[key] = np.arange(1, D, 2)
        with pytest.raises(ValueError, match="transforms dimensions"):
            spec_and_flat_from_state_dict(tampered)


def test_bench_flop_model_reproduces_the_survey_figures():
    """bench.py's analytic NSF FLOP count (used for workloads SURVEY.md 8(d) does not list) must give the survey's own
    figures for cfg3: 148 640 per draw + 30 000 per galaxy."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.nsf_flops(8, 20, 50, 5, 8) == (148640.0, 30000.0)
    w = b.WORKLOADS["nsf_cfg3"]
    assert (w["f_draw"], w["f_gal"], w["f_lp"]) == (148640.0, 30000.0, 178640.0)
    assert b.WORKLOADS["nsf_prod"]["f_lp"] == b.WORKLOADS["nsf_prod"]["f_draw"] + b.WORKLOADS["nsf_prod"]["f_gal"] > 9e5


def test_create_features_from_observations_follows_the_reference_rules():
    """ref: sbi_runner.py:2473-2937 -- column mapping and validation, the error-NaN check, the normalisation step as the
    reference writes it, rows with the missing-data flag removed, the magnitude-limit clip and inf -> NaN."""
    import pandas as pd
    from synference_amd import SBI_Fitter
    f = SBI_Fitter("m", ["a", "b"], feature_array=np.zeros((4, 6), np.float32),
                   feature_names=["F0", "F1", "unc_F0", "unc_F1", "norm_F2_AB", ], parameter_array=np.zeros((4, 2)))
    f.feature_array = np.zeros((4, 5), np.float32)
    with pytest.raises(ValueError, match="No feature array flags"):
        f.create_features_from_observations(pd.DataFrame({"F0": [1.0]}), flux_units="AB")
    f.feature_array_flags = dict(raw_observation_names=["F0", "F1"], error_names=["unc_F0", "unc_F1"],
                                 include_errors_in_feature_array=True, norm_name="norm_F2_AB", normalize_method="F2",
                                 normed_flux_units="AB", normalization_unit="AB", norm_mag_limit=30.0, remove_nan_inf=True)
    obs = pd.DataFrame({"m0": [24.0, 25.0, -99.0, 31.0, np.inf], "m1": [23.0, 26.0, 22.0, 27.0, 21.0],
                        "e0": [0.1, 0.2, 0.1, 0.1, 0.1], "e1": [0.1, 0.1, 0.1, 0.1, 0.1],
                        "nrm": [25.0, 24.0, 23.0, 26.0, 25.5]})
    cmap = {"m0": "F0", "m1": "F1", "e0": "unc_F0", "e1": "unc_F1", "nrm": "norm_F2_AB"}
    feat, removed = f.create_features_from_observations(obs, cmap, flux_units="AB")
    # the reference normalises BEFORE it looks for the missing-data flag (2844-2870): a flagged band of a normalised
    # model is no longer equal to the flag and the row stays -- reproduced as it is
    assert not removed.any() and feat.shape == (5, 5) and feat.dtype == np.float32
    nf = 10 ** ((23.9 - obs["nrm"].to_numpy()) / 2.5)                      # sbi_runner.py:2847
    want0 = (obs["m0"].to_numpy().astype(np.float32) - nf.astype(np.float32))
    exp0 = want0.copy()
    exp0[exp0 > 30.0] = 30.0                                               # norm_mag_limit clip (2889-2892): +inf too
    assert np.allclose(feat[:, 0], exp0, rtol=1e-6) and feat[4, 0] == 30.0
    assert np.allclose(feat[:, 1], obs["m1"].to_numpy().astype(np.float32) - nf.astype(np.float32), rtol=1e-6)
    assert np.allclose(feat[:, 2], obs["e0"].to_numpy()) and np.allclose(feat[:, 4], obs["nrm"].to_numpy())
    # without normalisation the flagged row goes, and faint magnitudes are clipped at the limit
    g = SBI_Fitter("m2", ["a", "b"], feature_array=np.zeros((4, 2), np.float32), feature_names=["F0", "F1"],
                   parameter_array=np.zeros((4, 2)))
    g.feature_array_flags = dict(raw_observation_names=["F0", "F1"], error_names=[], include_errors_in_feature_array=False,
                                 norm_name=None, normalize_method=None, normed_flux_units="AB", norm_mag_limit=30.0)
    feat_g, removed_g = g.create_features_from_observations(obs, {"m0": "F0", "m1": "F1"}, flux_units="AB")
    assert removed_g.tolist() == [False, False, True, False, False]
    assert np.allclose(feat_g[:, 0], [24.0, 25.0, 30.0, 30.0]) and np.allclose(feat_g[:, 1], [23.0, 26.0, 27.0, 21.0])
    obs_neg = obs.copy()
    obs_neg.loc[0, "m1"] = -np.inf                                         # -inf survives the clip and becomes NaN (2934)
    assert np.isnan(g.create_features_from_observations(obs_neg, {"m0": "F0", "m1": "F1"}, flux_units="AB")[0][0, 1])
    # validation messages
    with pytest.raises(AssertionError, match="do not match"):
        f.create_features_from_observations(obs, cmap, flux_units="nJy")
    with pytest.raises(ValueError, match="mapping for all photometry filters"):
        f.create_features_from_observations(obs, {k: v for k, v in cmap.items() if v != "F1"}, flux_units="AB")
    with pytest.raises(ValueError, match="mapping for all errors"):
        f.create_features_from_observations(obs, {k: v for k, v in cmap.items() if v != "unc_F0"}, flux_units="AB")
    with pytest.raises(ValueError, match="normalization factor"):
        f.create_features_from_observations(obs, {k: v for k, v in cmap.items() if v != "norm_F2_AB"}, flux_units="AB")
    bad = obs.copy()
    bad.loc[1, "e0"] = np.nan
    with pytest.raises(ValueError, match="contains NaN values where"):
        f.create_features_from_observations(bad, cmap, flux_units="AB")
    with pytest.raises(TypeError, match="pandas DataFrame"):
        f.create_features_from_observations(obs.to_numpy(), cmap, flux_units="AB")
    # NaN as the missing-data flag; ignore_missing keeps every row
    obs2 = obs.copy()
    obs2.loc[1, "m1"] = np.nan
    _, removed2 = g.create_features_from_observations(obs2, {"m0": "F0", "m1": "F1"}, flux_units="AB", missing_data_flag=np.nan)
    assert removed2.tolist() == [False, True, False, False, False]
    feat3, removed3 = g.create_features_from_observations(obs, {"m0": "F0", "m1": "F1"}, flux_units="AB", ignore_missing=True)
    assert not removed3.any() and feat3.shape == (5, 2)


# ------------------------------------------------------------------------------------------------
# round 3: rank-sharded catalogue evaluation (host side: row blocks, gather, seed agreement)
# ------------------------------------------------------------------------------------------------
def _shard_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synference_amd.posterior import all_gather_rows, broadcast_seed, dist_world, gather_rows, shard_bounds
    assert dist_world() == (rank, world)
    full = torch.arange(7 * 3 * 2, dtype=torch.float32).reshape(7, 3, 2)      # 7 rows over 2 ranks: blocks of 3 and 4
    b = shard_bounds(7, world)
    got = all_gather_rows(full[b[rank]:b[rank + 1]].clone(), b)
    seed = broadcast_seed(1234 if rank == 0 else 999)
    vec = all_gather_rows(torch.arange(b[rank], b[rank + 1], dtype=torch.float32), b)  # 1-D payload (log_prob)
    one = gather_rows(full[b[rank]:b[rank + 1]].clone(), b, dst=0)            # rank 0 only holds the whole array
    assert (one is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(one, full)
    ret[rank] = (got.clone(), seed, b, vec.clone())
    dist.destroy_process_group()


def test_row_sharding_gathers_uneven_blocks_and_agrees_on_the_seed():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(2, 33500 + (os.getpid() % 2000), ret), nprocs=2, join=True)
    full = torch.arange(7 * 3 * 2, dtype=torch.float32).reshape(7, 3, 2)
    for r in range(2):
        got, seed, b, vec = ret[r]
        assert b == [0, 3, 7]
        assert torch.equal(got, full) and seed == 1234
        assert torch.equal(vec, torch.arange(7, dtype=torch.float32))
    from synference_amd.posterior import shard_bounds
    for n, w in ((10, 3), (8, 8), (100001, 8)):
        b = shard_bounds(n, w)
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
        assert max(b[i + 1] - b[i] for i in range(w)) - min(b[i + 1] - b[i] for i in range(w)) <= 1


def test_oracle_streams_are_keyed_by_the_row_position():
    """oracle/posterior.py restates sf_flow_set_sample_row_offset: a block of rows with row_offset draws what those rows
    draw inside the whole catalogue."""
    from cases import make_case
    from oracle import posterior as OP
    ospec, spec, flat, theta, x = make_case("maf_small", B=6, spread=0.2)
    whole, _ = OP.sample(ospec, torch.as_tensor(flat), x, 9, 5, dtype=torch.float64)
    part, _ = OP.sample(ospec, torch.as_tensor(flat), x[2:5], 9, 5, dtype=torch.float64, row_offset=2)
    assert np.array_equal(whole[2:5], part)


# ------------------------------------------------------------------------------------------------
# round 3: the importer against a HAND-WRITTEN nflows state dict (real key names, mask / degree / permutation buffers)
# ------------------------------------------------------------------------------------------------
def test_importer_reads_a_hand_written_nflows_state_dict():
    """Every key below is spelled the way ``nflows.flows.Flow.state_dict()`` spells it for
    ``sbi.neural_nets.flow.build_maf`` (D = 2, H = 4, C = 1, one transform, z-scored theta and x): buffers included
    (``mask``, ``degrees``, ``_permutation``, ``_shape``, ``_log_z``).  Nothing here comes from ``param_layout`` or
    ``state_dict_from_flat``; the expected log-density is computed by the ~20 lines of numpy below, written from the
    nflows forward pass (MaskedLinear: y = (W * mask) x + b; no activation after the initial layer; tanh after each
    block's linear; ``unconstrained_scale, shift = out.view(-1, D, 2)``; ``scale = softplus(u) + 1e-3``)."""
    from oracle import flows as OF
    from synference_amd.importer import spec_and_flat_from_state_dict
    rs = np.random.RandomState(7)
    H, D, C = 4, 2, 1
    W0, b0 = rs.randn(H, D), rs.randn(H)
    Wc, bc = rs.randn(H, C), rs.randn(H)
    W1, b1, W2, b2 = rs.randn(H, H), rs.randn(H), rs.randn(H, H), rs.randn(H)
    Wf, bf = rs.randn(2 * D, H), rs.randn(2 * D)
    # nflows MADE buffers for features = 2, hidden = 4: input degrees [1, 2]; hidden degrees arange(4) % 1 + 1 = 1;
    # output degrees tile([1, 2], 2) in nflows' order (out_features = 2 * D, "repeat-interleave": [1, 1, 2, 2])
    m0 = np.array([[1., 0.]] * 4)                   # hidden_degree >= input_degree
    mh = np.ones((4, 4))
    mf = np.array([[0.] * 4, [0.] * 4, [1.] * 4, [1.] * 4])   # output_degree > hidden_degree
    mean_t, std_t = np.array([0.5, -1.0]), np.array([2.0, 0.5])
    mean_x, std_x = np.array([3.0]), np.array([4.0])
    pre = "_transform._transforms.1._transforms.0.autoregressive_net."
    sd = {
        "_transform._transforms.0._shift": (-mean_t / std_t).astype(np.float32),
        "_transform._transforms.0._scale": (1.0 / std_t).astype(np.float32),
        pre + "initial_layer.weight": W0.astype(np.float32), pre + "initial_layer.bias": b0.astype(np.float32),
        pre + "initial_layer.mask": m0.astype(np.float32), pre + "initial_layer.degrees": np.array([1, 1, 1, 1]),
        pre + "context_layer.weight": Wc.astype(np.float32), pre + "context_layer.bias": bc.astype(np.float32),
        pre + "blocks.0.linear.weight": W1.astype(np.float32), pre + "blocks.0.linear.bias": b1.astype(np.float32),
        pre + "blocks.0.linear.mask": mh.astype(np.float32), pre + "blocks.0.linear.degrees": np.array([1, 1, 1, 1]),
        pre + "blocks.1.linear.weight": W2.astype(np.float32), pre + "blocks.1.linear.bias": b2.astype(np.float32),
        pre + "blocks.1.linear.mask": mh.astype(np.float32), pre + "blocks.1.linear.degrees": np.array([1, 1, 1, 1]),
        pre + "final_layer.weight": Wf.astype(np.float32), pre + "final_layer.bias": bf.astype(np.float32),
        pre + "final_layer.mask": mf.astype(np.float32), pre + "final_layer.degrees": np.array([1, 1, 2, 2]),
        "_transform._transforms.1._transforms.1._permutation": np.array([1, 0]),
        "_embedding_net.0._mean": mean_x.astype(np.float32), "_embedding_net.0._std": std_x.astype(np.float32),
        "_distribution._shape": np.array([2]), "_distribution._log_z": np.array(0.5 * 2 * np.log(2 * np.pi)),
    }
    # a sbi >= 0.23 checkpoint carries the same keys behind "net."
    for prefix in ("", "net."):
        spec, flat = spec_and_flat_from_state_dict({prefix + k: v for k, v in sd.items()})
        assert (spec.kind, spec.D, spec.C, spec.H, spec.T, spec.NB) == ("maf", 2, 1, 4, 1, 2)
        assert list(spec.perms[0]) == [1, 0]
        ospec = OF.FlowSpec(kind="maf", D=2, C=1, H=4, T=1, perms=np.asarray(spec.perms),
                            theta_mean=np.asarray(spec.theta_mean, np.float64), theta_std=np.asarray(spec.theta_std, np.float64),
                            x_mean=np.asarray(spec.x_mean, np.float64), x_std=np.asarray(spec.x_std, np.float64))
        theta = rs.randn(9, 2) * std_t + mean_t
        x = rs.randn(9, 1) * std_x + mean_x
        got = OF.log_prob(ospec, torch.as_tensor(flat, dtype=torch.float64), torch.as_tensor(theta), torch.as_tensor(x)).numpy()
        # ---- the nflows forward pass, by hand (float32 weights as stored)
        f32 = lambda a: a.astype(np.float32).astype(np.float64)
        u = (theta - mean_t) / std_t
        e = (x - f32(mean_x)) / f32(std_x)
        h = u @ (f32(W0) * m0).T + f32(b0) + e @ f32(Wc).T + f32(bc)
        h = np.tanh(h @ (f32(W1) * mh).T + f32(b1))
        h = np.tanh(h @ (f32(W2) * mh).T + f32(b2))
        out = (h @ (f32(Wf) * mf).T + f32(bf)).reshape(-1, 2, 2)
        scale = np.logaddexp(0.0, out[..., 0]) + 1e-3
        v = scale * u + out[..., 1]
        z = v[:, [1, 0]]
        want = -0.5 * (z ** 2).sum(1) - np.log(2 * np.pi) + np.log(scale).sum(1) - np.log(std_t).sum()
        assert np.abs(got - want).max() < 1e-6, np.abs(got - want).max()
    # a checkpoint wired differently (other hidden degrees -> other masks) is refused, not silently re-masked
    bad = dict(sd)
    bad[pre + "initial_layer.mask"] = np.array([[1., 1.]] * 4, dtype=np.float32)
    with pytest.raises(ValueError, match="mask"):
        spec_and_flat_from_state_dict(bad)


def dense_order_slot(idx, M, S, G, run=1):
    """The block-interleaved dense order of a whole-catalogue sampling call, restated from sf_queue.h (sf_q_fetch, the
    `a.dense_G` branch; described at SfSampleArgsHost::dense_G / dense_run in sf_internal.h): item idx -> slot."""
    per_block = G * S
    b, j = divmod(idx, per_block)
    Gb = min(G, M - b * G)
    q, ln = divmod(j, run)
    rr, g = divmod(q, Gb)
    return (b * G + g) * S + rr * run + ln


@pytest.mark.parametrize("M,S,G", [(1, 7, 128), (5, 3, 128), (128, 4, 128), (129, 4, 128), (300, 5, 128), (2000, 3, 128),
                                   (37, 11, 8), (64, 2, 32), (70, 32, 16), (9, 48, 4)])
def test_block_interleaved_dense_order_is_a_permutation_that_spreads_a_galaxy(M, S, G):
    """Every slot exactly once (nothing sampled twice, nothing left empty), also with a ragged last block; and inside a
    full block a range of G consecutive items holds G different galaxies -- the point of the order."""
    slots = np.array([dense_order_slot(i, M, S, G) for i in range(M * S)])
    assert np.array_equal(np.sort(slots), np.arange(M * S))
    if M >= G:
        first = slots[:G] // S
        assert len(set(first.tolist())) == G
    # the slots of one galaxy are S items, one per run of Gb, never adjacent (unless the block is a single galaxy)
    gal = slots // S
    if M > 1:
        assert (np.diff(np.flatnonzero(gal == 0)) >= min(G, M)).all()
    # in runs of `run` draws (a tile of the kernel): still a permutation; a tile-aligned group of `run` items is `run`
    # consecutive slots of one galaxy
    for run in (r for r in (2, 4, 16) if S % r == 0):
        sl = np.array([dense_order_slot(i, M, S, G, run) for i in range(M * S)])
        assert np.array_equal(np.sort(sl), np.arange(M * S))
        grp = sl.reshape(-1, run)
        assert (np.diff(grp, axis=1) == 1).all() and (grp[:, 0] // S == grp[:, -1] // S).all()


@pytest.mark.parametrize("n", [4096, 4097, 5000, 8191, 8192, 100003])
def test_strided_walk_through_a_slot_list_is_a_permutation(n):
    """sf_queue.h (a.list_mul): item i of the dense list of an explicit slot list is entry walk(i) = i * mul mod 2^k,
    repeated until < n (cycle walking).  Every entry exactly once; neighbours far apart."""
    k = 12
    while (1 << k) < n:
        k += 1
    mul = ((1 << k) // 128) | 1
    msk = (1 << k) - 1
    pos = np.arange(n, dtype=np.uint64)
    out = np.empty(n, dtype=np.int64)
    todo = np.ones(n, bool)
    cur = pos.copy()
    while todo.any():
        cur[todo] = (cur[todo] * mul) & msk
        done = todo & (cur < n)
        out[done] = cur[done]
        todo &= ~done
    assert np.array_equal(np.sort(out), np.arange(n))
    assert np.median(np.abs(np.diff(out))) >= n // 256


# ------------------------------------------------------------------------------------------------
# round 4: the N = 8 launch of bench.py, without the workload (what the driver's 8-GPU run does before any kernel)
# ------------------------------------------------------------------------------------------------
def test_bench_spawns_eight_ranks_and_all_reduces_a_flat_gradient():
    """`python bench.py --gpus 8 --rendezvous-only` as ONE process: it spawns its eight ranks through torch.distributed.run on
    127.0.0.1 (before touching any GPU), every rank joins the process group (gloo here; RCCL on the node), all-reduces a flat
    gradient of the workload's size and rank 0 prints the OBSERVED world size and the all-reduce time; the children's exit
    code is the parent's."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SF_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--rendezvous-only", "--workload", "nsf_cfg3"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["rccl_ranks"] == 8 and rec["n_gpus"] == 8 and rec["backend"] == "gloo"
    assert rec["flat_gradient_floats"] == 91570 and rec["allreduce_sum_correct"]
    assert np.isfinite(rec["allreduce_us"]) and rec["allreduce_us"] > 0
    # a launcher that starts the wrong number of ranks is refused, never papered over
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--rendezvous-only"], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=4" in (r2.stdout + r2.stderr)


def test_bench_flop_model_of_the_autoregressive_nsf_counts_the_oracle_masks():
    """bench.py's nsfar_flops = 2 x the non-zeros of the oracle's masks (the context columns of the first layer separately:
    they are the same for every draw of a row), and D passes for the reference's inverse."""
    import importlib.util
    from oracle import flows as OF
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for D, C, H, T, K in ((5, 10, 50, 5, 8), (3, 4, 17, 2, 5), (8, 20, 64, 3, 8)):
        o = OF.FlowSpec(kind="nsf_ar", D=D, C=C, H=H, T=T, K=K, tail_bound=5.0)
        theta_nnz = ctx_nnz = 0
        for t in range(T):
            m0, m1, m2 = OF.ar_masks(o, t)
            theta_nnz += int(m0[:, :D].sum()) + int(m1.sum()) + int(m2.sum())
            ctx_nnz += int(m0[:, D:].sum())
        f_lp, f_draw, f_gal = b.nsfar_flops(D, C, H, T, K)
        assert f_lp == 2.0 * (theta_nnz + ctx_nnz) and f_gal == 2.0 * ctx_nnz and f_draw == 2.0 * D * theta_nnz
    w = b.WORKLOADS["nsfar_cfg2"]
    assert w["kind"] == "nsf_ar" and w["K"] == 8 and w["f_lp"] == b.nsfar_flops(5, 10, 50, 5, 8)[0]


def test_result_buffers_are_recycled_only_when_nobody_holds_them():
    """hostio.result_array: a large float64 result is handed out again once the caller has dropped it (and every view of it);
    a result that is still referenced is never touched; small arrays are not pooled."""
    from synference_amd import hostio
    shape = (300, 1000, 5)   # 12 MB
    a = hostio.result_array(shape)
    a[:] = 1.0
    ida = a.ctypes.data
    b = hostio.result_array(shape)          # a is still held: a different buffer
    assert b.ctypes.data != ida
    view = a[:10]
    del a
    c = hostio.result_array(shape)          # a view of a is still alive
    assert c.ctypes.data != ida and float(view[0, 0, 0]) == 1.0
    del view, c
    d = hostio.result_array((1000, 300, 5))  # same size, another shape: one of the dropped buffers comes back
    assert d.shape == (1000, 300, 5) and d.flags.c_contiguous and d.dtype == np.float64
    small = hostio.result_array((10, 10))
    assert small.nbytes < (8 << 20) and small.shape == (10, 10)
    os.environ["SF_HOSTIO_POOL"] = "0"
    try:
        e1 = hostio.result_array(shape); p1 = e1.ctypes.data; del e1
        e2 = hostio.result_array(shape)
        assert e2.shape == shape     # (pool off: a plain allocation each time -- nothing to assert about addresses)
    finally:
        del os.environ["SF_HOSTIO_POOL"]


def test_zuko_state_dict_importer_round_trip_and_connectivity_checks():
    """The lampe backend's flow (zuko.flows.NSF) in and out of the flat vector: module paths, mask / order buffers compared with
    the engine's own connectivity (and with the oracle's masks), masked weights dropped, wrong wiring refused."""
    from oracle import flows as OF
    from synference_amd.importer import spec_and_flat_from_zuko_state_dict, zuko_masks, zuko_state_dict_from_flat
    from synference_amd.spec import FlowSpec, init_params
    rng = np.random.default_rng(0)
    spec = FlowSpec(kind="nsf_ar", D=5, C=7, H=23, T=3, K=8, tail_bound=5.0, theta_mean=rng.normal(size=5), theta_std=rng.uniform(0.5, 2, 5),
                    x_mean=rng.normal(size=7), x_std=rng.uniform(0.5, 2, 7))
    flat = init_params(spec, torch.Generator().manual_seed(1)).numpy()
    sd = zuko_state_dict_from_flat(spec, flat, prefix="flow.")
    assert sd["flow.transform.transforms.1.order"].tolist() == [4, 3, 2, 1, 0] and sd["flow.transform.transforms.0.hyper.4.weight"].shape == (5 * 23, 23)
    o = OF.FlowSpec(kind="nsf_ar", D=5, C=7, H=23, T=3, K=8, tail_bound=5.0)
    for t in range(3):   # the product's masks are the oracle's
        for a, b in zip(zuko_masks(5, 7, 23, 23, OF.ar_order(o, t)), OF.ar_masks(o, t)):
            assert np.array_equal(a, b)
    spec2, flat2 = spec_and_flat_from_zuko_state_dict(sd, theta_mean=spec.theta_mean, theta_std=spec.theta_std, x_mean=spec.x_mean,
                                                      x_std=spec.x_std)
    assert (spec2.kind, spec2.D, spec2.C, spec2.H, spec2.T, spec2.K, spec2.tail_bound) == ("nsf_ar", 5, 7, 23, 3, 8, 5.0)
    # masked entries of the flat vector are dropped on the way (the kernels never read them), everything else survives
    lay = {n: (s, off) for n, s, off in OF.param_layout(o)}
    for t in range(3):
        for j, m in enumerate(OF.ar_masks(o, t)):
            s_, off = lay[f"t{t}.ar.W{j}"]
            n = int(np.prod(s_))
            assert np.array_equal(flat2[off:off + n].reshape(s_), flat[off:off + n].reshape(s_) * m)
            bs, boff = lay[f"t{t}.ar.b{j}"]
            assert np.array_equal(flat2[boff:boff + bs[0]], flat[boff:boff + bs[0]])
    bad = dict(sd)
    bad["flow.transform.transforms.2.hyper.2.mask"] = ~sd["flow.transform.transforms.2.hyper.2.mask"]
    with pytest.raises(ValueError, match="wired differently"):
        spec_and_flat_from_zuko_state_dict(bad)
    bad = dict(sd)
    bad["flow.transform.transforms.1.order"] = np.array([0, 1, 2, 3, 4])
    with pytest.raises(ValueError, match="alternating"):
        spec_and_flat_from_zuko_state_dict(bad)
    with pytest.raises(KeyError, match="not a zuko"):
        spec_and_flat_from_zuko_state_dict({"_transform._transforms.0._scale": np.ones(3)})
    three = dict(sd)
    three["flow.transform.transforms.0.hyper.6.weight"] = np.zeros((4, 23), np.float32)
    with pytest.raises(ValueError, match="hidden layers"):
        spec_and_flat_from_zuko_state_dict(three)

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

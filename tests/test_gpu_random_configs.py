"""Randomised configuration sweep (fixed seeds): every supported shape family against the fp64 oracle --
log_prob, inverse-from-noise, flat gradient and context gradient.  Catches layout / padding / staging
corner cases that the named cases do not hit (multi-tile contexts, HT = 1..4, D up to 16, K 2..16)."""
import numpy as np
import pytest
import torch

from oracle import flows as OF
from synference_amd.spec import FlowSpec

pytestmark = pytest.mark.gpu


def _configs():
    rng = np.random.default_rng(2026)
    out = []
    for i in range(28):
        kind = "maf" if i % 2 == 0 else "nsf"
        D = int(rng.integers(1 if kind == "maf" else 2, 17))
        C = int(rng.choice([1, 3, 8, 9, 31, 32, 33, 64, 70, 130]))
        H = int(rng.choice([4, 17, 32, 33, 50, 64, 65, 96, 100, 128]))
        T = int(rng.integers(1, 5))
        K = int(rng.integers(2, 17))
        NB = int(rng.choice([1, 2, 2, 2, 3, 4]))
        out.append((kind, D, C, H, T, K, NB, i))
    # corner cases by hand
    out += [("maf", 16, 512, 128, 1, 10, 2, 100), ("nsf", 16, 257, 128, 1, 16, 1, 101), ("nsf", 2, 1, 1, 3, 2, 2, 102),
            ("maf", 2, 1, 1, 2, 10, 2, 103), ("maf", 9, 20, 8, 2, 10, 2, 104)]
    return out


@pytest.mark.parametrize("kind,D,C,H,T,K,NB,seed", _configs())
def test_random_config_matches_oracle(kind, D, C, H, T, K, NB, seed):
    from synference_amd.engine import HipFlow
    rng = np.random.default_rng(seed)
    perms = OF.random_perms(D, T, seed) if kind == "maf" else None
    st = dict(theta_mean=rng.normal(size=D).astype(np.float32), theta_std=rng.uniform(0.5, 2, size=D).astype(np.float32),
              x_mean=rng.normal(size=C).astype(np.float32), x_std=rng.uniform(0.5, 2, size=C).astype(np.float32))
    ospec = OF.FlowSpec(kind=kind, D=D, C=C, H=H, T=T, K=K, NB=NB, perms=perms,
                        **{k: v.astype(np.float64) for k, v in st.items()})
    spec = FlowSpec(kind=kind, D=D, C=C, H=H, T=T, K=K, NB=NB, perms=perms, **st)
    flat = OF.init_params(ospec, seed + 1)
    flat = (flat + 0.4 * rng.normal(size=flat.shape) * np.abs(flat).mean()).astype(np.float32)
    B = 45
    theta = (rng.normal(size=(B, D)) * st["theta_std"] * 1.2 + st["theta_mean"]).astype(np.float32)
    x = (rng.normal(size=(B, C)) * st["x_std"] + st["x_mean"]).astype(np.float32)
    z = rng.normal(size=(B, D)).astype(np.float32)
    f = HipFlow(spec, "cuda:0")
    f.set_params(torch.as_tensor(flat))
    pt = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    lp_ref = OF.log_prob(ospec, pt, torch.as_tensor(theta).double(), xt)
    (-lp_ref).mean().backward()
    lp = f.log_prob(theta, x).cpu().double().numpy()
    assert np.abs(lp - lp_ref.detach().numpy()).max() < 1e-4 * max(1.0, D / 4)
    with torch.no_grad():
        th_ref, ld_ref = OF.inverse_transform(ospec, pt.detach(), torch.as_tensor(z).double(), xt.detach())
    th, ld = f.inverse(z, x)
    scale = np.maximum(np.abs(th_ref.numpy()), st["theta_std"])
    assert np.abs((th.cpu().double().numpy() - th_ref.numpy()) / scale).max() < 5e-4
    assert np.abs(ld.cpu().double().numpy() - ld_ref.numpy()).max() < 5e-4 * max(1.0, D / 4)
    dctx = torch.empty(B, C, device="cuda")
    _, grad = f.loss_grad(torch.as_tensor(flat), theta, x, 1.0 / B, dctx_out=dctx)
    g, rg = grad.cpu().double().numpy(), pt.grad.numpy()
    assert np.abs(g - rg).max() < 3e-4 * max(np.abs(rg).max(), 1e-6)
    rd = xt.grad.numpy()
    assert np.abs(dctx.cpu().double().numpy() - rd).max() < 3e-4 * max(np.abs(rd).max(), 1e-6)
    # sampler: finite, reproducible, and equal to inverse(Philox noise)
    s1 = f.sample(x[:3], 40, seed=5).cpu().numpy()
    s2 = f.sample(x[:3], 40, seed=5).cpu().numpy()
    assert np.isfinite(s1).all() and np.array_equal(s1, s2)


def _configs_other_kinds():
    """The flows outside the two tile engines: the autoregressive NSF of the lampe backend (csrc/sf_nsfar.hip: hidden rows sorted
    by type and padded, ragged H mod D, one parameter, 16 parameters, K = 2 .. 8) and sbi's one-parameter NSF (csrc/sf_nsf1.hip)."""
    rng = np.random.default_rng(2027)
    out = []
    for i in range(14):
        D = int(rng.integers(1, 17))
        C = int(rng.choice([1, 3, 8, 9, 31, 64, 70]))
        H = int(max(D, rng.choice([4, 17, 32, 33, 50, 64, 100, 128])))
        out.append(("nsf_ar", D, C, H, int(rng.integers(1, 5)), int(rng.integers(2, 9)), 200 + i))
    for i in range(6):
        out.append(("nsf", 1, int(rng.choice([1, 3, 8, 33, 70])), int(rng.choice([4, 17, 50, 64, 128])), int(rng.integers(1, 7)),
                    int(rng.integers(2, 17)), 300 + i))
    out += [("nsf_ar", 16, 64, 128, 1, 8, 400), ("nsf_ar", 2, 1, 2, 3, 2, 401), ("nsf_ar", 3, 5, 192, 1, 8, 402)]
    # zuko.flows.MAF (backend="lampe", model "maf"): the same engine with the affine univariate map
    out += [("maf_ar", 5, 10, 50, 5, 8, 500), ("maf_ar", 1, 3, 16, 2, 8, 501), ("maf_ar", 16, 31, 100, 2, 8, 502), ("maf_ar", 7, 9, 33, 3, 8, 503)]
    return out


@pytest.mark.parametrize("kind,D,C,H,T,K,seed", _configs_other_kinds())
def test_random_config_of_the_other_flow_kinds_matches_oracle(kind, D, C, H, T, K, seed):
    from synference_amd.engine import HipFlow
    rng = np.random.default_rng(seed)
    st = dict(theta_mean=rng.normal(size=D).astype(np.float32), theta_std=rng.uniform(0.5, 2, size=D).astype(np.float32),
              x_mean=rng.normal(size=C).astype(np.float32), x_std=rng.uniform(0.5, 2, size=C).astype(np.float32))
    extra = dict(tail_bound=5.0, ar_slope=float(rng.choice([1e-3, 1e-2]))) if kind in ("nsf_ar", "maf_ar") else {}
    ospec = OF.FlowSpec(kind=kind, D=D, C=C, H=H, T=T, K=K, **extra, **{k: v.astype(np.float64) for k, v in st.items()})
    spec = FlowSpec(kind=kind, D=D, C=C, H=H, T=T, K=K, **extra, **st)
    flat = OF.init_params(ospec, seed + 1)
    flat = (flat + 0.4 * rng.normal(size=flat.shape) * np.abs(flat).mean()).astype(np.float32)
    B = 77   # (two waves of the thread-per-sample kernels, the second ragged)
    theta = (rng.normal(size=(B, D)) * st["theta_std"] * 1.2 + st["theta_mean"]).astype(np.float32)
    x = (rng.normal(size=(B, C)) * st["x_std"] + st["x_mean"]).astype(np.float32)
    z = rng.normal(size=(B, D)).astype(np.float32)
    if kind in ("nsf_ar", "maf_ar"):   # a wave's rows must fit the CU's LDS: shapes beyond that are refused at creation, by name
        Hp = sum((len(range(r, H, D)) + 7) // 8 * 8 for r in range(D))
        Hp = (Hp + 15) // 16 * 16
        rows = (D + C + 15) // 16 * 16 + 2 * Hp + 2 * D          # (two hidden buffers; + the head rows of one wave)
        need = max((rows + 32) * 65 * 4, (rows + 24) * 65 * 4 + 4 * Hp * 4)   # density / sampling kernels, training kernel (+ its tables)
        if need > 160 * 1024 - 1024:
            with pytest.raises(RuntimeError, match="LDS"):
                HipFlow(spec, "cuda:0")
            return
    f = HipFlow(spec, "cuda:0")
    f.set_params(torch.as_tensor(flat))
    pt = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
    lp_ref = OF.log_prob(ospec, pt, torch.as_tensor(theta).double(), torch.as_tensor(x).double())
    (-lp_ref).mean().backward()
    lp = f.log_prob(theta, x).cpu().double().numpy()
    assert np.abs(lp - lp_ref.detach().numpy()).max() < 1e-4 * max(1.0, D / 4)
    with torch.no_grad():
        th_ref, ld_ref = OF.inverse_transform(ospec, pt.detach(), torch.as_tensor(z).double(), torch.as_tensor(x).double())
    th, ld = f.inverse(z, x)
    scale = np.maximum(np.abs(th_ref.numpy()), st["theta_std"])
    assert np.abs((th.cpu().double().numpy() - th_ref.numpy()) / scale).max() < 5e-4
    assert np.abs(ld.cpu().double().numpy() - ld_ref.numpy()).max() < 5e-4 * max(1.0, D / 4)
    loss, grad = f.loss_grad(torch.as_tensor(flat), theta, x, 1.0 / B)
    assert np.abs(loss.cpu().double().numpy() + lp_ref.detach().numpy()).max() < 1e-4 * max(1.0, D / 4)
    g, rg = grad.cpu().double().numpy(), pt.grad.numpy()
    assert np.abs(g - rg).max() < 3e-4 * max(np.abs(rg).max(), 1e-6)
    with pytest.raises(RuntimeError):
        f.loss_grad(torch.as_tensor(flat), theta, x, 1.0 / B, dctx_out=torch.empty(B, C, device="cuda"))
    s1 = f.sample(x[:3], 40, seed=5).cpu().numpy()
    s2 = f.sample(x[:3], 40, seed=5).cpu().numpy()
    assert np.isfinite(s1).all() and np.array_equal(s1, s2)

"""Pins for the CPU oracle that do not depend on any library (SURVEY.md 8c "what pins the build's
results instead"): analytic / self-consistency known answers.  The upstream stack is not importable
here and the reference ships no golden vectors for this path, so parity stays "unpinned" in the
formal sense; these tests are what holds the oracle in place."""
import math

import numpy as np
import pytest
import torch

from oracle import flows as OF
from oracle import philox
from oracle import posterior as OP


def _spec(kind, D, C, H=16, T=3, K=6, seed=0):
    perms = OF.random_perms(D, T, seed) if kind == "maf" else None
    return OF.FlowSpec(kind=kind, D=D, C=C, H=H, T=T, K=K, perms=perms)


def _rand_params(spec, seed=1, jitter=0.5):
    p = OF.init_params(spec, seed)
    rng = np.random.default_rng(seed)
    return torch.tensor(p + jitter * rng.normal(size=p.shape) * np.abs(p).mean())


@pytest.mark.parametrize("kind,D,C", [("maf", 5, 10), ("maf", 2, 3), ("nsf", 8, 20), ("nsf", 5, 4), ("nsf", 2, 3), ("nsf", 1, 4)])
def test_inverse_of_forward_is_identity_and_logdets_cancel(kind, D, C):
    spec = _spec(kind, D, C)
    p = _rand_params(spec)
    g = torch.Generator().manual_seed(0)
    th = torch.randn(17, D, generator=g, dtype=torch.float64) * 1.5
    x = torch.randn(17, C, generator=g, dtype=torch.float64)
    z, ld = OF.forward_transform(spec, p, th, x)
    th2, ld2 = OF.inverse_transform(spec, p, z, x)
    assert (th2 - th).abs().max() < 1e-10
    assert (ld + ld2).abs().max() < 1e-10


@pytest.mark.parametrize("kind,D,C", [("maf", 4, 3), ("nsf", 4, 3), ("nsf", 3, 2), ("nsf", 1, 2)])
def test_logdet_equals_autograd_jacobian(kind, D, C):
    spec = _spec(kind, D, C)
    p = _rand_params(spec)
    g = torch.Generator().manual_seed(1)
    th = torch.randn(3, D, generator=g, dtype=torch.float64)
    x = torch.randn(3, C, generator=g, dtype=torch.float64)
    _, ld = OF.forward_transform(spec, p, th, x)
    for i in range(3):
        J = torch.autograd.functional.jacobian(
            lambda t: OF.forward_transform(spec, p, t[None], x[i:i + 1])[0][0], th[i])
        assert abs(torch.linalg.slogdet(J)[1].item() - ld[i].item()) < 1e-10


def test_made_is_autoregressive_and_first_output_ignores_everything():
    """d out_i / d in_j = 0 for j >= i; output degree 1 is connected to no hidden unit."""
    D, C, H = 5, 3, 20
    spec = OF.FlowSpec(kind="maf", D=D, C=C, H=H, T=1)
    p = _rand_params(spec)
    P = OF.views(spec, p)
    masks = tuple(torch.tensor(m) for m in OF.made_masks(D, H))
    u = torch.randn(1, D, dtype=torch.float64)
    e = torch.randn(1, C, dtype=torch.float64)

    def f(uu):
        a, m = OF._made(spec, P, 0, uu[None], e, masks)
        return torch.cat([a[0], m[0]])

    J = torch.autograd.functional.jacobian(f, u[0])  # [2D, D]
    for i in range(D):
        for j in range(i, D):
            assert J[i, j].abs() < 1e-14 and J[D + i, j].abs() < 1e-14
    a0, m0 = OF._made(spec, P, 0, u, e, masks)
    a1, m1 = OF._made(spec, P, 0, u + 3.0, e * -2.0, masks)
    assert a0[0, 0] == a1[0, 0] == P["t0.bf"][0] and m0[0, 0] == m1[0, 0] == P["t0.bf"][1]
    deg_in, deg_h, deg_out = OF.made_degrees(D, H)
    assert deg_h.min() == 1 and deg_h.max() == D - 1 and list(deg_out[:4]) == [1, 1, 2, 2]


def test_maf_hand_computed_micro_case():
    """D=2, H=2, one transform, integer weights: log_prob by hand."""
    spec = OF.FlowSpec(kind="maf", D=2, C=1, H=2, T=1, NB=1)
    lay = {n: (s, o) for n, s, o in OF.param_layout(spec)}
    p = torch.zeros(OF.num_params(spec), dtype=torch.float64)

    def setp(name, val):
        s, o = lay[name]
        p[o:o + int(np.prod(s))] = torch.tensor(val, dtype=torch.float64).reshape(-1)

    setp("t0.W0", [[1.0, 5.0], [2.0, 7.0]])  # column 2 is masked (degree 2 feeds nothing)
    setp("t0.b0", [0.0, 0.0]); setp("t0.Wc", [[1.0], [0.0]]); setp("t0.bc", [0.0, 1.0])
    setp("t0.W1", [[1.0, 0.0], [0.0, 1.0]]); setp("t0.b1", [0.0, 0.0])
    setp("t0.Wf", [[9.0, 9.0], [9.0, 9.0], [1.0, 0.0], [0.0, 2.0]])  # rows 0,1 (dim 1) fully masked
    setp("t0.bf", [0.5, -1.0, 0.0, 0.25])
    th = torch.tensor([[0.3, -0.7]], dtype=torch.float64)
    x = torch.tensor([[2.0]], dtype=torch.float64)
    h0 = np.array([1 * 0.3 + 2.0, 2 * 0.3 + 1.0])
    h1 = np.tanh(h0)
    a = [0.5, 1.0 * h1[0]]
    m = [-1.0, 2.0 * h1[1] + 0.25]
    s = [math.log1p(math.exp(v)) + 1e-3 for v in a]
    z = [s[0] * 0.3 + m[0], s[1] * -0.7 + m[1]]
    expect = -0.5 * (z[0] ** 2 + z[1] ** 2) - math.log(2 * math.pi) + math.log(s[0]) + math.log(s[1])
    got = OF.log_prob(spec, p, th, x).item()
    assert abs(got - expect) < 1e-12


def test_spline_identity_at_init_tails_monotone_and_c1():
    spec = OF.FlowSpec(kind="nsf", D=2, C=1, H=8, T=1, K=5)
    # all-zero logits + the padded derivative constant everywhere -> the spline is the identity
    q = torch.zeros(1, 1, 3 * 5 - 1, dtype=torch.float64)
    q[..., 10:] = math.log(math.exp(1 - 1e-3) - 1)
    v = torch.linspace(-2.9, 2.9, 59, dtype=torch.float64)[:, None]
    out, lad = OF.rq_spline(spec, v, q.expand(59, 1, -1), inverse=False)
    assert (out - v).abs().max() < 1e-12 and lad.abs().max() < 1e-12
    # outside the tail bound: identity, logdet 0, for any parameters
    g = torch.Generator().manual_seed(3)
    qr = torch.randn(4, 1, 14, generator=g, dtype=torch.float64) * 3
    vo = torch.tensor([[-3.5], [3.0001], [10.0], [-3.0001]], dtype=torch.float64)
    out, lad = OF.rq_spline(spec, vo, qr, inverse=False)
    assert torch.equal(out, vo) and torch.equal(lad, torch.zeros_like(lad))
    # monotone, C1 across knots, logdet = log derivative, inverse exact
    qq = qr[:1].expand(2001, 1, 14)
    vv = torch.linspace(-3, 3, 2001, dtype=torch.float64)[:, None].clone().requires_grad_(True)
    out, lad = OF.rq_spline(spec, vv, qq, inverse=False)
    assert (out[1:] > out[:-1]).all()
    (d,) = torch.autograd.grad(out.sum(), vv)
    assert (torch.log(d) - lad).abs().max() < 1e-9
    # C1: the derivative just left and just right of every interior knot agrees, and equals 1 at +-B
    cw, _ = OF._knots(spec, qr[0, 0, :5] / math.sqrt(spec.H), spec.min_bin_width)
    kn = cw[1:-1]
    both = torch.cat([kn - 1e-7, kn + 1e-7, torch.tensor([-3 + 1e-9, 3 - 1e-9], dtype=torch.float64)])[:, None]
    both = both.clone().requires_grad_(True)
    o2, _ = OF.rq_spline(spec, both, qr[:1].expand(len(both), 1, 14), inverse=False)
    (d2,) = torch.autograd.grad(o2.sum(), both)
    assert (d2[:4] - d2[4:8]).abs().max() < 1e-4
    assert (d2[8:] - 1).abs().max() < 1e-6
    assert abs(out[0].item() + 3) < 1e-12 and abs(out[-1].item() - 3) < 1e-12
    back, lad2 = OF.rq_spline(spec, out.detach(), qq, inverse=True)
    assert (back - vv.detach()).abs().max() < 1e-9 and (lad2 + lad.detach()).abs().max() < 1e-9


def test_lu_identity_init_and_logdet():
    spec = OF.FlowSpec(kind="nsf", D=4, C=2, H=8, T=1, K=4)
    p = torch.tensor(OF.init_params(spec, 0))
    P = OF.views(spec, p)
    L, U, diag = OF._lu_mats(spec, P, 0)
    assert torch.allclose(L, torch.eye(4, dtype=torch.float64)) and (diag - 1).abs().max() < 1e-12
    assert torch.allclose(U, torch.eye(4, dtype=torch.float64))
    # index order: tril_indices / triu_indices row-major
    p2 = p.clone()
    lay = {n: o for n, s, o in OF.param_layout(spec)}
    p2[lay["t0.lu.lower"]:lay["t0.lu.lower"] + 6] = torch.arange(1, 7, dtype=torch.float64)
    p2[lay["t0.lu.upper"]:lay["t0.lu.upper"] + 6] = -torch.arange(1, 7, dtype=torch.float64)
    L, U, _ = OF._lu_mats(spec, OF.views(spec, p2), 0)
    assert L[1, 0] == 1 and L[2, 0] == 2 and L[2, 1] == 3 and L[3, 2] == 6
    assert U[0, 1] == -1 and U[0, 3] == -3 and U[1, 2] == -4 and U[2, 3] == -6


def test_density_integrates_to_one_2d():
    spec = _spec("nsf", 2, 2, H=8, T=2, K=4)
    p = _rand_params(spec, jitter=0.3)
    x = torch.tensor([[0.3, -0.4]], dtype=torch.float64)
    g = torch.linspace(-9, 9, 721, dtype=torch.float64)
    tt = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    lp = OF.log_prob(spec, p, tt, x.expand(len(tt), -1))
    integral = torch.exp(lp).sum().item() * (g[1] - g[0]).item() ** 2
    assert abs(integral - 1.0) < 2e-3


def test_standardize_stats_and_param_counts():
    rng = np.random.default_rng(0)
    th, x = rng.normal(size=(100, 3)) * [1, 10, 0], rng.normal(size=(100, 4))
    st = OF.standardize_stats(th, x)
    assert st["theta_std"][2] == pytest.approx(1e-14) and abs(st["theta_std"][1] - th[:, 1].std(ddof=1)) < 1e-5
    assert OF.num_params(OF.FlowSpec(kind="maf", D=5, C=10, H=50, T=5)) == 32300
    assert OF.num_params(OF.FlowSpec(kind="nsf", D=8, C=20, H=50, T=5, K=8)) == 91570


def test_philox_known_answers_random123():
    r = philox.philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(a[0]) for a in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = philox.philox4x32_10([0xffffffff], [0xffffffff], [0xffffffff], [0xffffffff], 0xffffffff, 0xffffffff)
    assert [int(a[0]) for a in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox.philox4x32_10([0x243f6a88], [0x85a308d3], [0x13198a2e], [0x03707344], 0xa4093822, 0x299f31d0)
    assert [int(a[0]) for a in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    z = philox.normal(5, np.arange(100000, dtype=np.uint64), 0, 3)
    assert np.abs(z.mean(0)).max() < 0.02 and np.abs(z.std(0) - 1).max() < 0.02


def test_rejection_sampler_and_ensemble_semantics():
    spec = _spec("maf", 3, 2, H=8, T=2)
    p = _rand_params(spec, jitter=0.2).float()
    x = np.random.default_rng(0).normal(size=(4, 2)).astype(np.float32)
    free, nd = OP.sample(spec, p, x, 300, 11)
    assert (nd == 300).all()
    lo, hi = np.quantile(free.reshape(-1, 3), 0.1, 0), np.quantile(free.reshape(-1, 3), 0.9, 0)
    s, nd = OP.sample(spec, p, x, 300, 11, lo, hi)
    assert OP.in_box(s.astype(np.float32), lo.astype(np.float32), hi.astype(np.float32)).all() and (nd > 300).all()
    keep = OP.in_box(free.astype(np.float32), lo.astype(np.float32), hi.astype(np.float32))
    assert np.array_equal(s[keep], free[keep])           # accepted first attempts are untouched
    acc = OP.acceptance(spec, p, x, 3000, 5, lo, hi)
    assert np.abs(acc - 300.0 / nd).max() < 0.08
    lp = OP.posterior_log_prob(spec, p, free[0, :5], np.repeat(x[:1], 5, 0), lo, hi)
    assert np.isneginf(lp[~keep[0, :5]]).all() and np.isfinite(lp[keep[0, :5]]).all()
    out = OP.ensemble_sample([spec, spec], [p, p * 1.01], [0.25, 0.75], x, 200, 3, lo, hi)
    assert np.isfinite(out).all()
    cnt = OP.ensemble_counts([0.25, 0.75], 200, 4, 3)
    assert (cnt.sum(1) == 200).all() and abs(cnt[:, 1].mean() / 200 - 0.75) < 0.1
    e = OP.ensemble_log_prob([spec, spec], [p, p], [0.3, 0.7], free[0, :5], np.repeat(x[:1], 5, 0))
    one = OP.posterior_log_prob(spec, p, free[0, :5], np.repeat(x[:1], 5, 0))
    assert np.abs(e - one).max() < 1e-9


# ---------------------------------------------------------------------------------------------------
# feature transforms (oracle/features.py): closed forms
# ---------------------------------------------------------------------------------------------------
def test_asinh_magnitude_limits_and_error():
    from oracle import features as F
    fb = np.array([5.0, 20.0])                                   # nJy
    # zero flux: mag = -2.5 log10(f_b / 3631 Jy)  (the softening magnitude; 5 nJy -> 29.65 AB)
    m0 = F.flux_to_asinh(np.zeros((1, 2)), fb)
    assert np.allclose(m0[0], -2.5 * np.log10(fb * 1e-9 / 3631.0), atol=1e-12)
    assert abs(m0[0, 0] - 29.65) < 0.01
    # bright limit: asinh magnitude -> AB magnitude
    f = np.array([[1e6, 3e7]])
    ab = -2.5 * np.log10(f * 1e-3) + 23.9
    assert np.allclose(F.flux_to_asinh(f, fb), ab, atol=2e-3)
    # negative flux is finite and fainter than the zero point, antisymmetric about it
    mp, mn = F.flux_to_asinh(np.array([[7.0, 7.0]]), fb), F.flux_to_asinh(np.array([[-7.0, -7.0]]), fb)
    assert np.allclose(mp + mn, 2 * m0, atol=1e-12) and (mn > m0).all()
    # error: d mag / d f * sigma
    f1 = np.array([[12.0, 40.0]]); s = np.array([[0.5, 2.0]])
    _, e = F.flux_to_asinh(f1, fb, s)
    h = 1e-4
    num = (F.flux_to_asinh(f1 + h, fb) - F.flux_to_asinh(f1 - h, fb)) / (2 * h)
    assert np.allclose(e, np.abs(num) * s, rtol=1e-6)


def test_scatter_depths_statistics_and_layout():
    from oracle import features as F
    f = np.tile(np.array([[100.0, 10.0, -3.0]]), (4000, 1))
    out, sig = F.scatter_depths(f, np.array([5.0, 10.0, 2.0]), n_scatters=3, depth_sigma=5.0, min_flux_pc_error=2.0, seed=9)
    assert out.shape == (12000, 3) and sig.shape == out.shape
    assert np.allclose(sig[0], [2.0, 2.0, 0.4])                   # max(depth/5, 2 % of |flux|)
    z = (out - np.repeat(f, 3, 0)) / sig
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    out2, _ = F.scatter_depths(f, np.array([5.0, 10.0, 2.0]), n_scatters=3, depth_sigma=5.0, min_flux_pc_error=2.0, seed=9)
    assert (out == out2).all()                                    # counter-based: reproducible


def test_pit_ranks_known_answer():
    from oracle import features as F
    s = np.arange(10, dtype=float).reshape(1, 10, 1).repeat(2, 2)
    s[0, 3, 1] = np.nan
    r = F.pit_ranks(s, np.array([[4.5, 4.5]]))
    assert np.allclose(r, [[0.5, 4 / 9]])
    assert np.isnan(F.pit_ranks(np.full((1, 4, 1), np.nan), np.zeros((1, 1)))).all()


def test_batched_accept_reject_fills_every_slot_inside_the_box():
    """oracle/posterior.py::accept_reject_sample_batched (the multi-core CPU baseline of bench.py): same target
    distribution as the per-galaxy loop -- every slot filled, inside the box, per-galaxy means close to the loop's."""
    import torch
    from cases import make_case
    from oracle import posterior as OP
    ospec, spec, flat, theta, x = make_case("maf_small", B=6, spread=0.2)
    fl = torch.as_tensor(flat)
    free, _ = OP.sample(ospec, fl, x, 300, 5, dtype=torch.float32)
    lo = np.quantile(free.reshape(-1, spec.D), 0.05, axis=0).astype(np.float32)
    hi = np.quantile(free.reshape(-1, spec.D), 0.95, axis=0).astype(np.float32)
    s, drawn = OP.accept_reject_sample_batched(ospec, fl, x, 400, lo, hi, torch.Generator().manual_seed(3))
    assert s.shape == (6, 400, spec.D) and np.isfinite(s).all() and drawn > 6 * 400
    assert ((s >= lo) & (s <= hi)).all()
    ref = np.stack([OP.accept_reject_sample(ospec, fl, x[g], 400, lo, hi, torch.Generator().manual_seed(10 + g))[0] for g in range(6)])
    sd = ref.std(1) + 1e-9
    assert (np.abs(s.mean(1) - ref.mean(1)) < 5 * sd / np.sqrt(400) * np.sqrt(2)).all()


def test_one_parameter_nsf_is_the_context_spline_map_flow():
    """[UPSTREAM] sbi build_nsf with a scalar theta (``x_numel == 1``): the single dimension is transformed in EVERY block by a
    spline whose parameters come from the context alone -- ContextSplineMap(hidden_layers=1) = Linear(C,H), ReLU, Linear(H,H),
    ReLU, Linear(H, 3K-1) -- and there is no LULinear.  Pins: parameter count, independence of the spline parameters from
    theta, the hand-evaluated conditioner, identity behaviour outside the tail bound, and a density that integrates to one."""
    D, C, H, T, K = 1, 3, 7, 4, 5
    spec = OF.FlowSpec(kind="nsf", D=D, C=C, H=H, T=T, K=K)
    assert spec.nsf_1d and not spec.has_lu
    per_t = H * C + H + H * H + H + (3 * K - 1) * H + (3 * K - 1)
    assert OF.num_params(spec) == T * per_t
    names = [n for n, _, _ in OF.param_layout(spec)]
    assert names[:6] == ["t0.csm.W0", "t0.csm.b0", "t0.csm.W1", "t0.csm.b1", "t0.csm.W2", "t0.csm.b2"] and not any("lu." in n for n in names)
    p = _rand_params(spec, seed=3, jitter=2.0)
    P = OF.views(spec, p)
    x = torch.tensor([[0.3, -1.2, 0.7]], dtype=torch.float64)
    # the conditioner by hand
    h = torch.relu(P["t1.csm.W0"] @ x[0] + P["t1.csm.b0"])
    h = torch.relu(P["t1.csm.W1"] @ h + P["t1.csm.b1"])
    q = P["t1.csm.W2"] @ h + P["t1.csm.b2"]
    assert torch.allclose(OF._context_spline_map(spec, P, 1, x)[0], q, atol=1e-14)
    # every block transforms the one dimension: a far-out theta passes through all of them unchanged (linear tails), an
    # interior one is moved, and the map is monotone
    far = torch.tensor([[25.0]], dtype=torch.float64)
    z, ld = OF.forward_transform(spec, p, far, x)
    assert abs(z.item() - 25.0) < 1e-12 and abs(ld.item()) < 1e-12
    grid = torch.linspace(-2.9, 2.9, 401, dtype=torch.float64)[:, None]
    zz, _ = OF.forward_transform(spec, p, grid, x.expand(len(grid), -1))
    assert (zz[1:] > zz[:-1]).all() and (zz - grid).abs().max() > 1e-3
    # integral of the density over theta
    g = torch.linspace(-14, 14, 400001, dtype=torch.float64)[:, None]
    lp = OF.log_prob(spec, p, g, x.expand(len(g), -1))
    assert abs(torch.trapezoid(lp.exp(), g[:, 0]).item() - 1.0) < 1e-6
    # z-scoring of the single parameter enters as an affine first layer
    spec2 = OF.FlowSpec(kind="nsf", D=1, C=C, H=H, T=T, K=K, theta_mean=np.array([2.0]), theta_std=np.array([0.5]))
    lp2 = OF.log_prob(spec2, p, torch.tensor([[2.3]], dtype=torch.float64), x)
    lp1 = OF.log_prob(spec, p, torch.tensor([[0.6]], dtype=torch.float64), x)
    assert abs((lp2 - lp1).item() - math.log(2.0)) < 1e-12


# ---- the autoregressive NSF of the lampe / zuko backend (oracle kind "nsf_ar") ------------------------------
def _ar_spec(D, C, H=12, T=3, K=5, **kw):
    return OF.FlowSpec(kind="nsf_ar", D=D, C=C, H=H, T=T, K=K, tail_bound=5.0, **kw)


@pytest.mark.parametrize("D,C", [(1, 3), (2, 3), (5, 4), (8, 2)])
def test_autoregressive_nsf_inverse_logdet_and_jacobian(D, C):
    spec = _ar_spec(D, C)
    p = _rand_params(spec)
    g = torch.Generator().manual_seed(0)
    th = torch.randn(9, D, generator=g, dtype=torch.float64) * 2.0
    x = torch.randn(9, C, generator=g, dtype=torch.float64)
    z, ld = OF.forward_transform(spec, p, th, x)
    th2, ld2 = OF.inverse_transform(spec, p, z, x)
    assert (th2 - th).abs().max() < 1e-10 and (ld + ld2).abs().max() < 1e-10
    for i in range(2):
        J = torch.autograd.functional.jacobian(lambda t: OF.forward_transform(spec, p, t[None], x[i:i + 1])[0][0], th[i])
        assert abs(torch.linalg.slogdet(J)[1].item() - ld[i].item()) < 1e-10


def test_autoregressive_nsf_masks_orders_and_triangular_jacobian():
    """zuko MaskedAutoregressiveTransform: transform t orders the dimensions 0..D-1 (t even) or D-1..0 (t odd); output i
    depends on input j only if order[j] < order[i] -- so ONE transform's Jacobian is triangular in that order with the spline
    derivative on the diagonal, the first-ordered dimension's spline parameters depend on the context alone, and every mask is
    the product rule type(out) >= type(in) (strict into the first hidden layer)."""
    D, C, H, K = 4, 3, 11, 5
    spec = _ar_spec(D, C, H=H, T=2, K=K)
    assert list(OF.ar_order(spec, 0)) == [0, 1, 2, 3] and list(OF.ar_order(spec, 1)) == [3, 2, 1, 0]
    m0, m1, m2 = OF.ar_masks(spec, 0)
    typ = np.arange(H) % D
    assert m0.shape == (H, D + C) and m1.shape == (H, H) and m2.shape == (D * (3 * K - 1), H)
    assert m0[:, D:].all() and not m0[typ == 0, :D].any() and m0[typ == 2, :2].all() and not m0[typ == 2, 2:D].any()
    assert (m1 == (typ[:, None] >= typ[None, :])).all()
    # connectivity of the mask product: parameter row of dimension i reaches input j iff order[j] < order[i]
    reach = (m2.astype(int) @ m1.astype(int) @ m0.astype(int))[:, :D] > 0
    for t in range(2):
        order = OF.ar_order(spec, t)
        mm = OF.ar_masks(spec, t)
        reach = (mm[2].astype(int) @ mm[1].astype(int) @ mm[0].astype(int))[:, :D] > 0
        for i in range(D):
            rows = reach[i * (3 * K - 1):(i + 1) * (3 * K - 1)]
            assert (rows == (order[None, :] < order[i])).all()
    one = _ar_spec(D, C, H=H, T=1, K=K)
    p = _rand_params(one)
    u = torch.randn(D, dtype=torch.float64)
    e = torch.randn(1, C, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(lambda t: OF.forward_transform(one, p, t[None], e)[0][0], u)
    assert torch.equal(J.triu(1), torch.zeros_like(J.triu(1))) and (torch.diagonal(J) > 0).all()
    P = OF.views(one, p)
    q_a = OF._ar_hyper(one, P, 0, u[None], e)
    q_b = OF._ar_hyper(one, P, 0, u[None] * -3.0 + 1.0, e)
    assert torch.equal(q_a[0, 0], q_b[0, 0]) and not torch.equal(q_a[0, 1], q_b[0, 1])


def test_zuko_spline_known_answers():
    """MonotonicRQSTransform: all-zero parameters are the identity on [-B, B] (equal bins, unit derivatives); outside the
    bound the map is the identity with log-derivative 0; logits are soft-clipped to +-|log slope| / 2 (widths, heights) and
    +-|log slope| (log-derivatives), so NO parameter value gives a bin narrower than 2B / (1 + (K - 1) / slope) or a knot
    derivative outside [slope, 1 / slope]; C1 at the knots and at +-B."""
    K, B, slope = 5, 5.0, 1e-3
    spec = _ar_spec(1, 1, K=K)
    v = torch.linspace(-4.9, 4.9, 50, dtype=torch.float64)[:, None]
    out, lad = OF.ar_spline(spec, v, torch.zeros(50, 1, 3 * K - 1, dtype=torch.float64), inverse=False)
    assert (out - v).abs().max() < 1e-12 and lad.abs().max() < 1e-12
    g = torch.Generator().manual_seed(2)
    q = torch.randn(1, 1, 3 * K - 1, generator=g, dtype=torch.float64) * 4
    vo = torch.tensor([[-5.5], [5.0001], [40.0]], dtype=torch.float64)
    out, lad = OF.ar_spline(spec, vo, q.expand(3, 1, -1), inverse=False)
    assert torch.equal(out, vo) and torch.equal(lad, torch.zeros_like(lad))
    huge = torch.zeros(1, 1, 3 * K - 1, dtype=torch.float64)
    huge[..., 0] = 1e9; huge[..., K] = -1e9; huge[..., 2 * K] = 1e9; huge[..., 2 * K + 1] = -1e9
    hor, ver, der = OF._ar_knots(spec, huge)
    ls = abs(math.log(slope))
    w = torch.diff(hor[0, 0]) / (2 * B)
    assert abs(w.max().item() - math.exp(ls / 2) / (math.exp(ls / 2) + (K - 1))) < 1e-6     # logit clipped to +ls/2, the others at 0
    hgt = torch.diff(ver[0, 0]) / (2 * B)
    assert abs(hgt.min().item() - math.exp(-ls / 2) / (math.exp(-ls / 2) + (K - 1))) < 1e-6
    assert abs(der[0, 0, 1].item() - 1 / slope) < 1e-3 / slope and abs(der[0, 0, 2].item() - slope) < 1e-3 * slope
    assert der[0, 0, 0] == 1 and der[0, 0, -1] == 1
    vv = torch.linspace(-5, 5, 2001, dtype=torch.float64)[:, None].clone().requires_grad_(True)
    qq = q.expand(2001, 1, -1)
    out, lad = OF.ar_spline(spec, vv, qq, inverse=False)
    (d,) = torch.autograd.grad(out.sum(), vv)
    assert (out[1:] > out[:-1]).all() and (torch.log(d) - lad).abs().max() < 1e-9
    hor, _, der = OF._ar_knots(spec, q)
    kn = hor[0, 0, 1:-1]
    both = torch.cat([kn - 1e-9, kn + 1e-9])[:, None].clone().requires_grad_(True)
    o2, _ = OF.ar_spline(spec, both, q.expand(len(both), 1, -1), inverse=False)
    (d2,) = torch.autograd.grad(o2.sum(), both)
    knot_d = der[0, 0, 1:-1]
    assert ((d2[:K - 1, 0] - d2[K - 1:, 0]).abs() / knot_d).max() < 1e-4 and ((d2[:K - 1, 0] - knot_d).abs() / knot_d).max() < 1e-4
    back, lad2 = OF.ar_spline(spec, out.detach(), qq, inverse=True)
    assert (back - vv.detach()).abs().max() < 1e-9 and (lad2 + lad.detach()).abs().max() < 1e-9


def test_autoregressive_nsf_density_integrates_to_one_2d():
    spec = _ar_spec(2, 2, H=8, T=2, K=4)
    p = _rand_params(spec, jitter=0.3)
    x = torch.tensor([[0.3, -0.4]], dtype=torch.float64)
    g = torch.linspace(-10, 10, 801, dtype=torch.float64)
    tt = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1).reshape(-1, 2)
    lp = OF.log_prob(spec, p, tt, x.expand(len(tt), -1))
    assert abs(torch.exp(lp).sum().item() * (g[1] - g[0]).item() ** 2 - 1.0) < 2e-3


# ---- the MAF of the lampe / zuko backend (oracle kind "maf_ar": zuko.flows.MAF) -------------------------------------------------
@pytest.mark.parametrize("D,C", [(1, 3), (3, 2), (6, 4)])
def test_zuko_maf_inverse_logdet_jacobian_and_affine_known_answers(D, C):
    """zuko.flows.MAF = the same MaskedAutoregressiveTransform with MonotonicAffineTransform(shift, scale, slope): the scale logit
    is soft-clipped -- log_scale = s / (1 + |s / log slope|), so |log_scale| < |log slope| whatever the network says --, zero
    parameters are the identity, a transform's Jacobian is triangular in its order with exp(log_scale) on the diagonal, the
    inverse undoes the forward pass to 1e-10 and the log-determinants agree with autograd's."""
    spec = OF.FlowSpec(kind="maf_ar", D=D, C=C, H=12, T=3, K=8, tail_bound=5.0)
    assert spec.ar_np == 2 and OF.num_params(spec) == 3 * (12 * (D + C) + 12 + 12 * 12 + 12 + 2 * D * 12 + 2 * D)
    p = _rand_params(spec)
    g = torch.Generator().manual_seed(0)
    th = torch.randn(9, D, generator=g, dtype=torch.float64) * 2.0
    x = torch.randn(9, C, generator=g, dtype=torch.float64)
    z, ld = OF.forward_transform(spec, p, th, x)
    th2, ld2 = OF.inverse_transform(spec, p, z, x)
    assert (th2 - th).abs().max() < 1e-10 and (ld + ld2).abs().max() < 1e-10
    for i in range(2):
        J = torch.autograd.functional.jacobian(lambda t: OF.forward_transform(spec, p, t[None], x[i:i + 1])[0][0], th[i])
        assert abs(torch.linalg.slogdet(J)[1].item() - ld[i].item()) < 1e-10
    v = torch.linspace(-3, 3, 7, dtype=torch.float64)[:, None]
    out, lad = OF.ar_affine(spec, v, torch.zeros(7, 1, 2, dtype=torch.float64), inverse=False)
    assert torch.equal(out, v) and torch.equal(lad, torch.zeros_like(lad))
    q = torch.tensor([[[0.7, 1e9]], [[-0.2, -1e9]]], dtype=torch.float64)
    out, lad = OF.ar_affine(spec, torch.ones(2, 1, dtype=torch.float64), q, inverse=False)
    ls = abs(math.log(spec.ar_slope))
    assert abs(lad[0, 0].item() - ls) < 1e-6 and abs(lad[1, 0].item() + ls) < 1e-6        # clipped to +-|log slope|
    assert abs(out[0, 0].item() - (math.exp(lad[0, 0].item()) + 0.7)) < 1e-9
    one = OF.FlowSpec(kind="maf_ar", D=max(D, 2), C=C, H=12, T=1, K=8)
    p1 = _rand_params(one)
    u = torch.randn(one.D, dtype=torch.float64)
    e = torch.randn(1, C, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(lambda t: OF.forward_transform(one, p1, t[None], e)[0][0], u)
    assert torch.equal(J.triu(1), torch.zeros_like(J.triu(1))) and (torch.diagonal(J) > 0).all()


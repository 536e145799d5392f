"""Randomised shapes of the rejection sampler against the oracle's per-slot schedule.

The named sampler tests fix the catalogue shape; the queue of the persistent kernel has shape-dependent corners --
catalogues smaller than one workgroup iteration (64 items), a single galaxy or a single draw, slot counts just around
the iteration size and the ticket-ring mode, attempt ceilings on the edges of the launch windows (64, 256, 1024),
several flows per process.  Every case: all rows filled or NaN exactly where the oracle has NaN, draws inside the box,
the oracle's draws up to boundary flips, attempt counts equal up to those flips.
"""
import numpy as np
import pytest
import torch

from cases import make_case
from oracle import posterior as OP
from synference_amd.engine import HipFlow

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(77)
    out = []
    shapes = [(1, 1), (1, 7), (5, 1), (1, 64), (2, 31), (3, 21), (63, 1), (65, 1), (4, 16), (9, 57), (1, 1000), (130, 3)]
    flows = ["maf_small", "maf_cfg1", "nsf_nb1", "maf_span6"]
    for i, (M, S) in enumerate(shapes):
        name = flows[i % len(flows)]
        q = float(rng.choice([0.02, 0.1, 0.2, 0.28]))           # box = central (1 - 2q) quantile range per dimension
        cap = [None, None, 1, 3, 64, 65, 257, 1025][int(rng.integers(0, 8))]
        out.append((name, M, S, q, cap, 1000 + i))
    return out


@pytest.mark.parametrize("name,M,S,q,cap,seed", _cases())
def test_random_catalogue_shapes_and_ceilings_match_the_oracle(name, M, S, q, cap, seed):
    ospec, spec, flat, theta, x = make_case(name, B=M, spread=0.2)
    free, _ = OP.sample(ospec, torch.as_tensor(flat), x[:min(M, 16)], 200, 5, dtype=torch.float32)
    free = free.reshape(-1, ospec.D)
    lo = np.quantile(free, q, axis=0).astype(np.float32)
    hi = np.quantile(free, 1.0 - q, axis=0).astype(np.float32)
    f = HipFlow(spec, "cuda:0")
    f.set_params(torch.as_tensor(flat))
    f.set_sample_time_limit(60.0)                                # a ceiling far above anything these sizes need
    got, nd = f.sample(x, S, lo, hi, seed=seed, max_attempts=cap, return_counts=True)
    got, nd = got.cpu().double().numpy(), nd.cpu().numpy()
    ref, rnd = OP.sample(ospec, torch.as_tensor(flat), x, S, seed, lo, hi, max_attempts=cap, dtype=torch.float32)
    assert got.shape == ref.shape == (M, S, spec.D)
    nan_g, nan_r = np.isnan(got).all(-1), np.isnan(ref).all(-1)
    assert np.isnan(got).any(-1).sum() == nan_g.sum()            # a row is either complete or all NaN
    assert f.last_unfilled == int(nan_g.sum())
    if cap is None:
        assert nan_g.sum() == 0
    # accept / reject flips at the box boundary can move a slot to another attempt: rare, and bounded here
    both = ~nan_g & ~nan_r
    assert (nan_g != nan_r).mean() <= 0.05 + 1.0 / (M * S)
    fin = got[both]
    assert ((fin >= lo) & (fin <= hi)).all()
    err = np.abs((fin - ref[both]) / (hi - lo).astype(np.float64)).max(-1) if both.any() else np.zeros(1)
    assert (err > 5e-4).mean() <= 0.05 + 1.0 / max(1, both.sum()), ((err > 5e-4).mean(), err.max())
    assert np.abs(nd - rnd).sum() <= 0.08 * rnd.sum() + 2


def test_many_handles_and_back_to_back_catalogues_share_the_device_cleanly():
    """Two flows alive at once, catalogues of different shapes in turn on each: handle-owned queues and scratch are
    re-sized per shape and never leak state from one call into the next."""
    a = make_case("maf_cfg1", B=40, spread=0.2)
    b = make_case("nsf_nb1", B=17, spread=0.2)
    fa, fb = HipFlow(a[1], "cuda:0"), HipFlow(b[1], "cuda:0")
    fa.set_params(torch.as_tensor(a[2]))
    fb.set_params(torch.as_tensor(b[2]))
    outs = {}
    for rep in range(2):
        for tag, (f, case, M, S) in {"a_big": (fa, a, 40, 300), "b": (fb, b, 17, 50), "a_small": (fa, a, 3, 5)}.items():
            x = case[4][:M]
            got = f.sample(x, S, seed=9).cpu().numpy()           # no box: one attempt per slot
            assert np.isfinite(got).all()
            if rep == 0:
                outs[tag] = got
            else:
                assert np.array_equal(outs[tag], got)            # same seed, same slots -> same draws, whatever ran in between


@pytest.mark.parametrize("name,M,S", [("maf_cfg1", 300, 400), ("nsf_nb1", 150, 200)])
def test_draws_do_not_depend_on_the_dense_order(name, M, S):
    """The whole-catalogue call walks its dense list across the galaxies in blocks (sf_internal.h, dense_G); streams are
    keyed by slot and attempt and a slot keeps its LOWEST accepted attempt, so the output must be bit-identical in plain
    slot order, with the default block and with a ragged small block -- three fresh processes (the order is an
    environment switch read once per process)."""
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(__file__), "helpers", "order_child.py")
    digests = []
    # ... and without the per-galaxy context table (size cap 0: the kernels evaluate the context products per draw, on the
    # dispatching kernels instead of the unrolled ones): the table holds the same MFMA sums
    for extra in ({"SF_INTERLEAVE": "0"}, {"SF_INTERLEAVE": "128"}, {"SF_INTERLEAVE": "7"}, {"SF_CTAB_MAX_MB": "0"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, child, name, str(M), str(S)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1]
        digests.append(line)
    assert digests[0] == digests[1] == digests[2], digests
    # without the table the products are summed in another order (NSF: the table's context products; MAF since round 5: the table
    # path runs the FUSED first layer W' u + c0', the no-table path the two layers): equal to fp32 rounding, not bit for bit
    assert abs(float(digests[3].split()[3]) - float(digests[0].split()[3])) <= 1e-6 * abs(float(digests[0].split()[3]))
    assert digests[0].split()[2] == "0"   # every slot filled


@pytest.mark.parametrize("D,H,T,NB", [(5, 64, 2, 1), (5, 40, 3, 2), (5, 52, 1, 2), (4, 48, 2, 1), (4, 30, 4, 2), (3, 32, 2, 1),
                                      (3, 20, 5, 2), (5, 50, 5, 1)])
def test_unrolled_sampler_shapes_match_the_oracle(D, H, T, NB):
    """One degree group per 16-row tile (H <= 16 (D - 1)), D = 3 ... 5, one or two blocks: the shapes of the unrolled
    sampler kernels (k_maf_samp16<NB, .., DD>, compile-time image offsets) and of the deep tail's k_maf_find16s."""
    from oracle import flows as OF
    from synference_amd.spec import FlowSpec
    rng = np.random.default_rng(1000 * D + H + T)
    C = int(rng.integers(3, 20))
    perms = OF.random_perms(D, T, 7 + H)
    st = dict(theta_mean=rng.normal(size=D).astype(np.float32), theta_std=rng.uniform(0.5, 2, size=D).astype(np.float32),
              x_mean=rng.normal(size=C).astype(np.float32), x_std=rng.uniform(0.5, 2, size=C).astype(np.float32))
    ospec = OF.FlowSpec(kind="maf", D=D, C=C, H=H, T=T, K=10, NB=NB, perms=perms, **{k: v.astype(np.float64) for k, v in st.items()})
    spec = FlowSpec(kind="maf", D=D, C=C, H=H, T=T, K=10, NB=NB, perms=perms, **st)
    flat = OF.init_params(ospec, 11 + H)
    flat = (flat + 0.3 * rng.normal(size=flat.shape) * np.abs(flat).mean()).astype(np.float32)
    M, S = 7, 160
    x = (rng.normal(size=(M, C)) * st["x_std"] + st["x_mean"]).astype(np.float32)
    f = HipFlow(spec, "cuda:0")
    d = f.describe()
    assert d["m16_ok"] and not d["m16_span"] and d["nT16"] == D - 1, d     # really one of the unrolled shapes
    f.set_params(torch.as_tensor(flat))
    free, _ = OP.sample(ospec, torch.as_tensor(flat), x, 300, 99, dtype=torch.float32)
    lo = np.quantile(free.reshape(-1, D), 0.1, axis=0).astype(np.float32)
    hi = np.quantile(free.reshape(-1, D), 0.9, axis=0).astype(np.float32)
    got, nd = f.sample(x, S, lo, hi, seed=3, return_counts=True)
    got, nd = got.cpu().double().numpy(), nd.cpu().numpy()
    ref, rnd = OP.sample(ospec, torch.as_tensor(flat), x, S, 3, lo, hi, dtype=torch.float32)
    assert f.last_unfilled == 0 and np.isfinite(got).all() and ((got >= lo) & (got <= hi)).all()
    err = np.abs((got - ref) / (hi - lo).astype(np.float64)).max(-1)
    assert (err > 5e-4).mean() < 0.01, ((err > 5e-4).mean(), err.max())
    assert np.abs(nd - rnd).sum() <= max(3, 0.02 * rnd.sum())


def test_later_persistent_windows_equal_the_find_and_resolve_launches():
    """Slots that use up the first 1 024 attempts go to find / resolve launches, or -- while many are open (>= 8 192; here
    forced with SF_PERSIST_MIN=1) -- to further persistent launches over the windows [1 024, 16 384) ...: the same draws
    and attempt counts, bit for bit (the find / resolve route is checked against the oracle in test_gpu_sampler.py)."""
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(__file__), "helpers", "order_child.py")
    lines = []
    for extra in ({}, {"SF_PERSIST_MIN": "1"}):
        r = subprocess.run([sys.executable, child, "maf_cfg1", "2", "24", "0.36"], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1].split())
    assert lines[0][1] == lines[1][1] and lines[0][2] == lines[1][2] == "0", lines
    assert int(lines[0][4]) >= 3 and int(lines[1][4]) == 2, lines     # find / resolve pairs vs a second persistent launch

"""CPU checks of the cooperative 16-row training image (csrc/sf_layout.cpp, SfTrcDev; kernel: csrc/sf_trainc.hip).

A numpy model replays the kernel's TILE algebra (blocks of 16 x 16 taken from the image with the kernel's lane
formulas, input-tile rows, head-tile rows, slot conventions) in float64 and must reproduce the oracle's log_prob;
the transposed blocks must be the transposes of the forward blocks, and every logical parameter must map to the
gradient-partial position of the accumulator element that multiplies it."""
import numpy as np
import pytest
import torch

from cases import make_case, oracle_log_prob

NAMES = ["maf_cfg1", "maf_small", "maf_sig2", "maf_span6", "maf_span_h64", "maf_d2_span"]


def fwd_block(img, off, nb, a, b):
    """16 x 16 matrix W[ro][ri] of forward block (a, b): lane l, comp r = W[l & 15][4 * (l >> 4) + r]."""
    blk = img[off + (a * nb + b) * 256: off + (a * nb + b + 1) * 256].reshape(64, 4)
    W = np.zeros((16, 16))
    for l in range(64):
        for r in range(4):
            W[l & 15, 4 * (l >> 4) + r] = blk[l, r]
    return W


def tr_block(img, off, nb, a, b):
    """transposed block (it = a, ot = b): lane l, comp r = W[ot rows 4 * (l >> 4) + r][it rows l & 15] -> returns [ro][ri]."""
    blk = img[off + (a * nb + b) * 256: off + (a * nb + b + 1) * 256].reshape(64, 4)
    W = np.zeros((16, 16))
    for l in range(64):
        for r in range(4):
            W[4 * (l >> 4) + r, l & 15] = blk[l, r]
    return W


@pytest.mark.parametrize("name", NAMES)
def test_trainc_image_reproduces_log_prob_and_gradient_map(name):
    from synference_amd.engine import HipFlow
    ospec, spec, flat, theta, x = make_case(name, B=40)
    hf = HipFlow(spec)
    tab = hf.trainc_table()
    assert tab is not None, "shape should have a cooperative training image"
    s1, s2, gd, d, cst = tab
    fl = np.concatenate([flat.astype(np.float64), [0.0]])
    img = fl[s1] + fl[s2]          # (-1 -> the appended zero)
    NT, NI, D, C, T = d["NT"], d["NI"], spec.D, spec.C, spec.T
    desc = hf.describe()
    c_ps, c_sh, c_td = 0, 16, 32
    c_xm = 48
    c_xs = 48 + ((C + 3) // 4) * 4
    insrc = cst[d["c_insrc"]: d["c_insrc"] + NI * 16].astype(int)
    B = len(theta)
    # ---- inputs
    tdim = cst[c_td: c_td + D].astype(int)
    u = np.zeros((8, B))
    for p in range(D):
        u[p] = theta[:, tdim[p]] * cst[c_ps + p] + cst[c_sh + p]
    xs = (x.astype(np.float64) - cst[c_xm: c_xm + C]) / cst[c_xs: c_xs + C]
    inx = np.zeros((NI, 16, B))
    for it in range(NI):
        for rho in range(16):
            f = insrc[it * 16 + rho]
            if f >= 0:
                inx[it, rho] = xs[:, f]
    slot_rows = [(4 * (p >> 1) + (p & 1)) for p in range(D)]
    assert all(insrc[r] < 0 for r in slot_rows)
    assert sorted(q for q in insrc if q >= 0) == list(range(C))
    logdet = np.zeros(B)
    kend = [d[f"kend{q}"] for q in range(4)]
    kbeg = [d[f"kbeg{q}"] for q in range(4)]
    for t in range(T):
        tb = t * d["t_stride"]
        inp = inx.copy()
        for p in range(D):
            inp[0, slot_rows[p]] = u[p]
        h0 = [img[tb + d["o_b0"] + ot * 16: tb + d["o_b0"] + ot * 16 + 16][:, None]
              + sum(fwd_block(img, tb + d["o_win"], NI, ot, it) @ inp[it] for it in range(NI)) for ot in range(NT)]
        a1 = [np.tanh(img[tb + d["o_b1"] + ot * 16: tb + d["o_b1"] + ot * 16 + 16][:, None]
                      + sum(fwd_block(img, tb + d["o_w1"], NT, ot, it) @ h0[it] for it in range(kend[ot] + 1))) for ot in range(NT)]
        a2 = [np.tanh(img[tb + d["o_b2"] + ot * 16: tb + d["o_b2"] + ot * 16 + 16][:, None]
                      + sum(fwd_block(img, tb + d["o_w2"], NT, ot, it) @ a1[it] for it in range(kend[ot] + 1))) for ot in range(NT)]
        # blocks beyond kend must be structurally zero (the kernel skips them)
        for ot in range(NT):
            for it in range(kend[ot] + 1, NT):
                assert not fwd_block(img, tb + d["o_w1"], NT, ot, it).any()
                assert not fwd_block(img, tb + d["o_w2"], NT, ot, it).any()
            for it in range(NT):
                if ot < kbeg[it]:
                    assert not fwd_block(img, tb + d["o_w1"], NT, ot, it).any()
        fin = img[tb + d["o_bf"]: tb + d["o_bf"] + 16][:, None] + sum(fwd_block(img, tb + d["o_wf"], NT, 0, it) @ a2[it] for it in range(NT))
        for p in range(D):
            g4, r = p >> 1, p & 1
            av, mv = fin[4 * g4 + r], fin[4 * g4 + 2 + r]
            if ospec.scale_fn == "softplus":
                sc = np.logaddexp(0, av) + ospec.maf_eps
            else:
                sc = 1 / (1 + np.exp(-(av + 2))) + ospec.maf_eps
            u[p] = sc * u[p] + mv
            logdet += np.log(sc)
        # transposed blocks are the transposes
        for a_ in range(NT):
            for b_ in range(NT):
                np.testing.assert_array_equal(tr_block(img, tb + d["o_w2T"], NT, a_, b_), fwd_block(img, tb + d["o_w2"], NT, b_, a_))
                np.testing.assert_array_equal(tr_block(img, tb + d["o_w1T"], NT, a_, b_), fwd_block(img, tb + d["o_w1"], NT, b_, a_))
            np.testing.assert_array_equal(tr_block(img, tb + d["o_wfT"], 1, a_, 0), fwd_block(img, tb + d["o_wf"], NT, 0, a_))
            for it in range(NI):
                np.testing.assert_array_equal(tr_block(img, tb + d["o_winT"], NT, it, a_), fwd_block(img, tb + d["o_win"], NI, a_, it))
        # gradient map: the accumulator element of (block, ro, ri) belongs to the logical parameter packed there
        for (o_key, g_key, OT, IT) in (("o_win", "g_win", NT, NI), ("o_w1", "g_w1", NT, NT), ("o_w2", "g_w2", NT, NT), ("o_wf", "g_wf", 1, NT)):
            for ot in range(OT):
                for it in range(IT):
                    base = tb + d[o_key] + (ot * IT + it) * 256
                    for l in range(64):
                        for r in range(4):
                            src = s1[base + l * 4 + r]
                            if src >= 0:
                                ro, ri = l & 15, 4 * (l >> 4) + r
                                want = t * d["g_stride"] + d[g_key] + (ot * IT + it) * 256 + (ro & 3) * 64 + (ro >> 2) * 16 + ri
                                assert gd[src] == want
        for (o_key, g_key, OT) in (("o_b0", "g_b0", NT), ("o_b1", "g_b1", NT), ("o_b2", "g_b2", NT), ("o_bf", "g_bf", 1)):
            for q in range(OT * 16):
                for src in (s1[tb + d[o_key] + q], s2[tb + d[o_key] + q]):
                    if src >= 0:
                        assert gd[src] == t * d["g_stride"] + d[g_key] + q
    logdet0 = -np.log(np.asarray(ospec.theta_std, dtype=np.float64)).sum()
    lp = -0.5 * (u[:D] ** 2).sum(0) - 0.5 * D * np.log(2 * np.pi) + logdet + logdet0
    ref = oracle_log_prob(ospec, flat, theta, x)
    assert np.abs(lp - ref).max() < 1e-6, np.abs(lp - ref).max()  # (the constants image is float32)
    # every unmasked parameter has a gradient slot; masked ones have none
    p = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
    from oracle import flows as OF
    OF.log_prob(ospec, p, torch.as_tensor(theta).double(), torch.as_tensor(x).double()).sum().backward()
    nz = p.grad.numpy() != 0
    assert (gd[nz] >= 0).all()
    assert gd.max() < d["n_grad"]


def test_shapes_without_a_cooperative_image():
    from synference_amd.engine import HipFlow
    for name in ("maf_wide", "maf_nb3", "maf_d1", "nsf_nb1", "nsf_k16"):
        _, spec, *_ = make_case(name, B=4)
        assert HipFlow(spec).trainc_table() is None

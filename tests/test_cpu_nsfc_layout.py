"""CPU checks of the cooperative 16-row NSF training image (csrc/sf_layout.cpp, SfNscDev; kernel: csrc/sf_nsfc.hip).

A numpy model replays the kernel's TILE algebra in float64 -- 16 x 16 blocks taken from the image with the kernel's lane
formulas, its input-tile rows, hidden-row placement, spline-head slots (lane (sample, row group g) owns the parameters of
transformed dimension g), the padded LU block -- and must reproduce the oracle's log_prob; the transposed blocks must be
the transposes of the forward blocks, and every logical parameter must map to the gradient position of the accumulator
element that multiplies it (including the two LU blocks and their row sums)."""
import numpy as np
import pytest
import torch

from cases import make_case, oracle_log_prob
from oracle import flows as OF
from test_cpu_trainc_layout import fwd_block, tr_block

NAMES = ["nsf_cfg3", "nsf_odd", "nsf_k10", "nsf_d2", "nsf_h69"]


def _case(name, B):
    return make_case(name, B=B)


@pytest.mark.parametrize("name", NAMES)
def test_nsfc_image_reproduces_log_prob_and_gradient_map(name):
    from synference_amd.engine import HipFlow
    ospec, spec, flat, theta, x = _case(name, 24)
    hf = HipFlow(spec)
    tab = hf.trainc_table()
    assert tab is not None and hf.describe()["nsfc_ok"] == 1
    s1, s2, gd, d, cst = tab
    assert (s2 < 0).all()
    fl = np.concatenate([flat.astype(np.float64), [0.0]])
    img = fl[s1]
    NT, NI, OTQ, KM = d["NT"], d["NI"], d["OTQ"], d["KM"]
    D, C, T, K, H = spec.D, spec.C, spec.T, spec.K, spec.H
    B = len(theta)
    assert NT == -(-H // 16) and NI == -(-(8 + C) // 16) and OTQ * 4 >= 3 * K - 1 and 3 * KM - 1 <= 4 * OTQ
    assert d["kc_h"] == -(-(H - 16 * (NT - 1)) // 4)
    c_ps, c_sh, c_xm = 0, 16, 48
    c_xs = 48 + ((C + 3) // 4) * 4
    u = np.zeros((8, B))
    for p in range(D):
        u[p] = theta[:, p] * cst[c_ps + p] + cst[c_sh + p]
    xs = (x.astype(np.float64) - cst[c_xm: c_xm + C]) / cst[c_xs: c_xs + C]
    inx = np.zeros((NI, 16, B))
    seen = []
    for it in range(NI):
        for rho in range(16):
            g, m = rho >> 2, rho & 3
            if it == 0 and m < 2:
                continue
            f = (m - 2) * 4 + g if it == 0 else 8 + (it - 1) * 16 + 4 * m + g
            if f < C:
                inx[it, rho] = xs[:, f]
                seen.append(f)
                if it >= 1:   # the components the kernel multiplies cover every used row
                    assert m < d[f"kc_in{it}"]
    assert sorted(seen) == list(range(C))
    ospec_t = ospec
    logdet = np.zeros(B)
    q_of = lambda k: "01"[k]
    for t in range(T):
        tb = t * d["t_stride"]
        start = t & 1
        d_tr = (D - start + 1) // 2
        inp = inx.copy()
        for p in range(D):
            inp[0, 4 * (p >> 1) + (p & 1)] = u[p]
        bias = lambda key, ot: img[tb + d[key] + ot * 16: tb + d[key] + ot * 16 + 16][:, None]
        h = [bias("o_bin", ot) + sum(fwd_block(img, tb + d["o_win"], NI, ot, it) @ inp[it] for it in range(NI)) for ot in range(NT)]
        # rows of the last hidden tile beyond kc_h components are padding: their weights must be zero everywhere
        for k in range(2):
            kk = q_of(k)
            r = [np.maximum(v, 0) for v in h]
            t1 = [bias("o_b1" + kk, ot) + sum(fwd_block(img, tb + d["o_w1" + kk], NT, ot, it) @ r[it] for it in range(NT)) for ot in range(NT)]
            r = [np.maximum(v, 0) for v in t1]
            t2 = [bias("o_b2" + kk, ot) + sum(fwd_block(img, tb + d["o_w2" + kk], NT, ot, it) @ r[it] for it in range(NT)) for ot in range(NT)]
            gate = [bias("o_bg" + kk, ot) + sum(fwd_block(img, tb + d["o_wg" + kk], NI, ot, it) @ inp[it] for it in range(NI)) for ot in range(NT)]
            # the gate product skips components 0 and 1 of input tile 0 (theta rows): those weights must be zero
            for ot in range(NT):
                W = fwd_block(img, tb + d["o_wg" + kk], NI, ot, 0)
                assert not W[:, [4 * g + m for g in range(4) for m in range(2)]].any()
            h = [h[ot] + t2[ot] / (1 + np.exp(-gate[ot])) for ot in range(NT)]
            last = fwd_block(img, tb + d["o_w1" + kk], NT, 0, NT - 1)
            assert not last[:, [4 * g + m for g in range(4) for m in range(d["kc_h"], 4)]].any()
        qt = [bias("o_bout", ot) + sum(fwd_block(img, tb + d["o_wout"], NT, ot, it) @ h[it] for it in range(NT)) for ot in range(OTQ)]
        # lane (sample, g) collects slots 4 j + r of dimension g from tile j
        q_slots = np.zeros((4, 4 * OTQ, B))
        for jt in range(OTQ):
            for rho in range(16):
                q_slots[rho >> 2, 4 * jt + (rho & 3)] = qt[jt][rho]
        vin = np.stack([u[start + 2 * g] for g in range(d_tr)], 1)                       # [B, d_tr]
        qq = np.zeros((B, d_tr, 3 * K - 1))
        for g in range(d_tr):
            qq[:, g, :K] = q_slots[g, :K].T
            qq[:, g, K:2 * K] = q_slots[g, KM:KM + K].T
            qq[:, g, 2 * K:] = q_slots[g, 2 * KM:2 * KM + K - 1].T
        vout, lad = OF.rq_spline(ospec_t, torch.as_tensor(vin), torch.as_tensor(qq), False)
        for g in range(d_tr):
            u[start + 2 * g] = vout[:, g].numpy()
        logdet += lad.sum(1).numpy()
        # LULinear from the padded block
        lu = img[tb + d["o_lu"]: tb + d["o_lu"] + 144]
        Lm, Um, ud, bb = lu[:64].reshape(8, 8), lu[64:128].reshape(8, 8), lu[128:136], lu[136:144]
        assert not np.triu(Lm).any() and not np.tril(Um).any()
        dg = np.where(np.arange(8) < D, np.logaddexp(0, ud) + ospec.lu_eps, 1.0)
        tt = dg[:, None] * u + Um @ u
        u = tt + Lm @ tt + bb[:, None]
        u[D:] = 0
        logdet += np.log(dg[:D]).sum()
        # transposed blocks are the transposes
        for a_ in range(NT):
            for b_ in range(NT):
                for k in range(2):
                    np.testing.assert_array_equal(tr_block(img, tb + d[f"o_w2T{k}"], NT, a_, b_), fwd_block(img, tb + d[f"o_w2{k}"], NT, b_, a_))
                    np.testing.assert_array_equal(tr_block(img, tb + d[f"o_w1T{k}"], NT, a_, b_), fwd_block(img, tb + d[f"o_w1{k}"], NT, b_, a_))
            for ot in range(OTQ):
                np.testing.assert_array_equal(tr_block(img, tb + d["o_woutT"], OTQ, a_, ot), fwd_block(img, tb + d["o_wout"], NT, ot, a_))
            np.testing.assert_array_equal(tr_block(img, tb + d["o_winT"], NT, 0, a_), fwd_block(img, tb + d["o_win"], NI, a_, 0))
        # gradient map of the weight blocks and biases
        layers = [("o_win", "g_win", NT, NI), ("o_wout", "g_wout", OTQ, NT)]
        for k in range(2):
            layers += [(f"o_wg{k}", f"g_wg{k}", NT, NI), (f"o_w1{k}", f"g_w1{k}", NT, NT), (f"o_w2{k}", f"g_w2{k}", NT, NT)]
        for (o_key, g_key, OT, IT) in layers:
            for ot in range(OT):
                for it in range(IT):
                    base = tb + d[o_key] + (ot * IT + it) * 256
                    for l in range(64):
                        for r in range(4):
                            src = s1[base + l * 4 + r]
                            if src >= 0:
                                ro, ri = l & 15, 4 * (l >> 4) + r
                                assert gd[src] == t * d["g_stride"] + d[g_key] + (ot * IT + it) * 256 + (ro & 3) * 64 + (ro >> 2) * 16 + ri
        biases = [("o_bin", "g_bin", NT), ("o_bout", "g_bout", OTQ)]
        for k in range(2):
            biases += [(f"o_bg{k}", f"g_bg{k}", NT), (f"o_b1{k}", f"g_b1{k}", NT), (f"o_b2{k}", f"g_b2{k}", NT)]
        for (o_key, g_key, OT) in biases:
            for qi in range(OT * 16):
                src = s1[tb + d[o_key] + qi]
                if src >= 0:
                    assert gd[src] == t * d["g_stride"] + d[g_key] + qi
        # LU: block 0 element (i, j) = dL[i][j], block 1 = dU[i][j], then rows [0, 8) dbias and [8, 16) d udiag
        gl = t * d["g_stride"] + d["g_lu"]
        pos = lambda blk, ro, ri: gl + blk * 256 + (ro & 3) * 64 + (ro >> 2) * 16 + ri
        for i in range(8):
            for jj in range(8):
                sL, sU = s1[tb + d["o_lu"] + i * 8 + jj], s1[tb + d["o_lu"] + 64 + i * 8 + jj]
                assert (sL >= 0) == (jj < i < D) and (sU >= 0) == (i < jj < D)
                if sL >= 0:
                    assert gd[sL] == pos(0, i, jj)
                if sU >= 0:
                    assert gd[sU] == pos(1, i, jj)
            sd, sb = s1[tb + d["o_lu"] + 128 + i], s1[tb + d["o_lu"] + 136 + i]
            assert (sd >= 0) == (i < D) and (sb >= 0) == (i < D)
            if i < D:
                assert gd[sd] == gl + 512 + 8 + i and gd[sb] == gl + 512 + i
    logdet0 = -np.log(np.asarray(ospec.theta_std, dtype=np.float64)).sum()
    lp = -0.5 * (u[:D] ** 2).sum(0) - 0.5 * D * np.log(2 * np.pi) + logdet + logdet0
    ref = oracle_log_prob(ospec, flat, theta, x)
    assert np.abs(lp - ref).max() < 1e-5, np.abs(lp - ref).max()   # (the constants image is float32; the spline amplifies it)
    # every parameter has exactly one gradient position, no two share one
    assert (gd >= 0).all() and gd.max() < d["n_grad"] and len(np.unique(gd)) == len(gd)


def test_shapes_without_a_cooperative_nsf_image():
    from synference_amd.engine import HipFlow
    for name in ("nsf_nb1", "nsf_k16"):   # one residual block; 47 spline parameters per dimension
        _, spec, *_ = make_case(name, B=4)
        assert HipFlow(spec).describe()["nsfc_ok"] == 0

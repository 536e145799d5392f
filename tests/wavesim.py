"""numpy model of ONE wave of the HIP kernels (test infrastructure, CPU only).

Replays the data flow of synference_amd/csrc/sf_flows.h -- MFMA 32x32x2 lane maps, the packed
operand image, bias image, u / context tiles, slot maps -- in float64, so that the host-side
packer (sf_layout.cpp) and the layout conventions can be checked against the oracle without a
GPU.  It mirrors the kernel's indexing, not its arithmetic shortcuts.
"""
from __future__ import annotations

import numpy as np

LANES = np.arange(64)
C_ = LANES & 31
H_ = LANES >> 5


def row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma(a_lane, b_lane, acc):
    """v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col=l&31,row=row(r,l>>5)."""
    A = a_lane.reshape(2, 32).T
    Bm = b_lane.reshape(2, 32)
    Dm = A @ Bm
    for r in range(16):
        acc[r] += Dm[row(r, H_), C_]


def init_bias(packed, off, OT):
    acc = np.zeros((OT, 16, 64))
    for mt in range(OT):
        for r in range(16):
            acc[mt, r] = packed[off + (mt * 2 + H_) * 16 + r]
    return acc


def mm_acc(acc, tiles, packed, woff, nGtot, kg0, ng, relu=False):
    OT = acc.shape[0]
    for mt in range(OT):
        for g in range(ng):
            base = woff + ((mt * nGtot + kg0 + g) * 64 + LANES) * 4
            for j in range(4):
                b = tiles[g >> 2][(g & 3) * 4 + j]
                if relu:
                    b = np.maximum(b, 0)
                mfma(packed[base + j], b, acc[mt])


def u_tile(u):
    t = np.zeros((16, 64))
    for r in range(8):
        t[r] = np.where(H_ == 1, u[row(r, 1)], u[row(r, 0)])
    return t


def ctx_tiles(x_rows, d, cst):
    """x_rows [64 lanes, C] raw context of each lane's sample."""
    Cc = d["C"]
    out = []
    for kt in range((d["nGc"] + 3) // 4):
        t = np.zeros((16, 64))
        for r in range(16):
            rho = kt * 32 + row(r, H_)
            ok = rho < Cc
            rr = np.where(ok, rho, 0)
            v = (x_rows[LANES, rr] - cst[d["c_xmean"] + rr]) / cst[d["c_xstd"] + rr]
            t[r] = np.where(ok, v, 0.0)
        out.append(t)
    return out


def ctx_mm(acc, ct, packed, woff, d):
    for kt, t in enumerate(ct):
        ng = min(4, d["nGc"] - kt * 4)
        mm_acc(acc, [t], packed, woff, d["nGc"], kt * 4, ng)


def xhalf(v):
    return v[LANES ^ 32]


def softplus(x):
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def maf_logprob(d, packed, theta32, x32):
    """theta32 [32,D], x32 [32,C] -> log_prob [32] for one 32-sample tile."""
    cst = np.asarray(d["cst"])
    D, HT = d["D"], d["HT"]
    th = theta32[C_]
    xr = x32[C_]
    u = np.zeros((16, 64))
    for p in range(D):
        td = int(cst[d["c_tdim"] + p])
        u[p] = th[:, td] * cst[d["c_pscale"] + p] + cst[d["c_pshift"] + p]
    logdet = np.full(64, np.sum(np.log(np.abs(cst[d["c_pscale"]:d["c_pscale"] + D]))))
    ct = ctx_tiles(xr, d, cst)
    for t in range(d["T"]):
        tp = t * d["t_stride"]
        a = init_bias(packed, tp + d["o_b0"], HT)
        mm_acc(a, [u_tile(u)], packed, tp + d["o_w0"], d["nGu"], 0, d["nGu"])
        ctx_mm(a, ct, packed, tp + d["o_wc"], d)
        for k in range(d["NB"]):
            b = init_bias(packed, tp + d[f"o_bk{k}"], HT)
            mm_acc(b, list(a), packed, tp + d[f"o_wk{k}"], d["nGh"], 0, d["nGh"])
            a = np.tanh(b)
        fin = init_bias(packed, tp + d["o_bf"], 1)
        mm_acc(fin, list(a), packed, tp + d["o_wf"], d["nGh"], 0, d["nGh"])
        ld = np.zeros(64)
        for p in range(D):
            a_ = fin[0, 2 * (p >> 1)]
            s = (softplus(a_) if d.get("scale_fn", 0) == 0 else 1 / (1 + np.exp(-(a_ + 2.0)))) + 1e-3
            val = s * u[p] + fin[0, 2 * (p >> 1) + 1]
            mine = H_ == (p & 1)
            u[p] = np.where(mine, val, xhalf(val))
            ld += np.where(mine, np.log(s), 0.0)
        logdet += ld + xhalf(ld)
    lp = -0.5 * (u[:D] ** 2).sum(0) - 0.5 * D * np.log(2 * np.pi) + logdet
    return lp[:32]


def _spline_fwd(d, q, v):
    """q: [PT*16 slots, 64 lanes] ; v [64] -> (out, lad) mirroring SfSpline::eval (forward)."""
    K, KM, B = d["K"], d["KMAX"], 3.0
    H = d["H"]

    def knots(off, min_size):
        e = q[off:off + K] / np.sqrt(H)
        e = np.exp(e - e.max(0))
        w = min_size + (1 - min_size * K) * e / e.sum(0)
        cs = np.cumsum(w, 0)
        c = np.concatenate([np.full((1, 64), -B), 2 * B * cs[:-1] - B, np.full((1, 64), B)], 0)
        return c

    inside = (v >= -B) & (v <= B)
    vc = np.clip(v, -B, B)
    cw = knots(0, 1e-3)
    ch = knots(KM, 1e-3)
    idx = np.zeros(64, int)
    for k in range(K):
        idx = np.where(vc >= cw[k], k, idx)
    L = LANES
    x_k, w_k = cw[idx, L], cw[idx + 1, L] - cw[idx, L]
    y_k, h_k = ch[idx, L], ch[idx + 1, L] - ch[idx, L]
    const = np.log(np.exp(1 - 1e-3) - 1)
    der = np.concatenate([np.full((1, 64), const), q[2 * KM:2 * KM + K - 1], np.full((1, 64), const)], 0)
    der = 1e-3 + softplus(der)
    d_k, d_k1 = der[idx, L], der[idx + 1, L]
    s_k = h_k / w_k
    xi = (vc - x_k) / w_k
    om = xi * (1 - xi)
    den = s_k + (d_k + d_k1 - 2 * s_k) * om
    out = y_k + h_k * (s_k * xi * xi + d_k * om) / den
    dnum = s_k * s_k * (d_k1 * xi * xi + 2 * s_k * om + d_k * (1 - xi) ** 2)
    lad = np.log(dnum) - 2 * np.log(den)
    return np.where(inside, out, v), np.where(inside, lad, 0.0)


def nsf_logprob(d, packed, theta32, x32):
    cst = np.asarray(d["cst"])
    D, HT, PT = d["D"], d["HT"], d["PT"]
    th = theta32[C_]
    xr = x32[C_]
    u = np.zeros((16, 64))
    for p in range(D):
        u[p] = th[:, p] * cst[d["c_pscale"] + p] + cst[d["c_pshift"] + p]
    logdet = np.full(64, np.sum(np.log(np.abs(cst[d["c_pscale"]:d["c_pscale"] + D]))))
    ct = ctx_tiles(xr, d, cst)
    sig = lambda z: 1 / (1 + np.exp(-z))
    for t in range(d["T"]):
        tp = t * d["t_stride"]
        hid = init_bias(packed, tp + d["o_bin"], HT)
        mm_acc(hid, [u_tile(u)], packed, tp + d["o_winu"], d["nGu"], 0, d["nGu"])
        ctx_mm(hid, ct, packed, tp + d["o_winc"], d)
        for k in range(d["NB"]):
            t1 = init_bias(packed, tp + d[f"o_b1{k}"], HT)
            mm_acc(t1, list(hid), packed, tp + d[f"o_w1{k}"], d["nGh"], 0, d["nGh"], relu=True)
            t2 = init_bias(packed, tp + d[f"o_b2{k}"], HT)
            mm_acc(t2, list(t1), packed, tp + d[f"o_w2{k}"], d["nGh"], 0, d["nGh"], relu=True)
            for mt in range(HT):
                g = init_bias(packed, tp + d[f"o_bg{k}"] + mt * 32, 1)
                ctx_mm(g, ct, packed, tp + d[f"o_wg{k}"] + mt * d["nGc"] * 256, d)
                hid[mt] += t2[mt] * sig(g[0])
        start = t & 1
        d_tr = (D - start + 1) // 2
        for jp in range((d_tr + 1) // 2):
            q = init_bias(packed, tp + d["o_bout"] + jp * PT * 32, PT)
            mm_acc(q, list(hid), packed, tp + d["o_wout"] + jp * PT * d["nGh"] * 256, d["nGh"], 0, d["nGh"])
            qs = q.reshape(PT * 16, 64)
            kdim = 2 * jp + H_
            have = kdim < d_tr
            tgt = start + 2 * kdim
            tgt_o = start + 2 * (2 * jp + (1 - H_))
            have_o = (2 * jp + (1 - H_)) < d_tr
            vin = u[np.minimum(tgt, 15), LANES]
            vout, lad = _spline_fwd(d, qs, vin)
            lad = np.where(have, lad, 0.0)
            vo = xhalf(vout)
            for p in range(16):
                u[p] = np.where(have & (p == tgt), vout, u[p])
                u[p] = np.where(have_o & (p == tgt_o), vo, u[p])
            logdet += lad + xhalf(lad)
        if D > 1:
            lp_ = tp + d["o_lu"]
            Lm = packed[lp_:lp_ + D * D].reshape(D, D)
            Um = packed[lp_ + D * D:lp_ + 2 * D * D].reshape(D, D)
            ud = packed[lp_ + 2 * D * D:lp_ + 2 * D * D + D]
            bb = packed[lp_ + 2 * D * D + D:lp_ + 2 * D * D + 2 * D]
            dg = softplus(ud) + 1e-3
            tt = np.zeros((D, 64))
            for i in range(D):
                tt[i] = dg[i] * u[i] + sum(Um[i, j] * u[j] for j in range(i + 1, D))
            for i in range(D):
                u[i] = tt[i] + bb[i] + sum(Lm[i, j] * tt[j] for j in range(i))
            logdet += np.log(dg).sum()
    lp = -0.5 * (u[:D] ** 2).sum(0) - 0.5 * D * np.log(2 * np.pi) + logdet
    return lp[:32]


TANH_PRESCALE = float(np.float32(2.8853900817779268))   # SF_TANH_PRESCALE (sf_layout.h)


def pack(flat, s1, s2):
    """The pack kernels' rule (k_pack): flat[s1] + flat[s2], -1 = nothing; second index -2 (SF_PACK_TANH_SCALE: hidden
    blocks of the 16-row MAF images) = the value times 2 log2(e)."""
    f = np.concatenate([np.asarray(flat, dtype=np.float64), [0.0]])
    s2 = np.asarray(s2)
    v = f[s1] + f[np.where(s2 == -2, -1, s2)]  # index -1 hits the appended zero
    return np.where(s2 == -2, v * TANH_PRESCALE, v)


# ---------------------------------------------------------------------------------------------------
# 16-row engine (synference_amd/csrc/sf_maf16.hip): v_mfma_f32_16x16x4_f32 lane maps, the packed16 image and the
# incremental autoregressive inverse, one 16-draw tile
# ---------------------------------------------------------------------------------------------------
S16 = LANES & 15
G4 = LANES >> 4


def mfma16(a_lane, b_lane, acc4, r_unused=None):
    """v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: lane (j=l&15, g=l>>4), reg r = row 4g+r."""
    A = a_lane.reshape(4, 16).T
    Bm = b_lane.reshape(4, 16)
    Dm = A @ Bm
    for r in range(4):
        acc4[r] += Dm[4 * G4 + r, S16]


def mma16(packed, woff, IT, ot, it, tile_in, acc4):
    base = woff + ((ot * IT + it) * 64 + LANES) * 4
    for r in range(4):
        mfma16(packed[base + r], tile_in[r], acc4)


def maf_inverse16(d, packed, z16, x16, head_mfma=False):
    """z16 [16,D] base noise, x16 [16,C] context rows -> theta [16,D] through the 16-row image (table-free path).
    head_mfma: the (a, m) head rows as ONE MFMA output tile accumulated over the finished hidden tiles (o16_wh / o16_bh,
    the HM variant of sf_pass16b) instead of the per-lane dot products over o16_hv."""
    cst = np.asarray(d["cst"])
    D, T, NB, NT, C = d["D"], d["T"], d["NB"], d["nT16"], d["C"]
    # draw in tile layout: lane (s, g4), reg r = physical slot 4*g4 + r
    u = np.zeros((4, 64))
    for r in range(4):
        p = 4 * G4 + r
        u[r] = np.where(p < D, z16[S16, np.minimum(p, D - 1)], 0.0)
    # standardised context tiles
    def ctx_tile(ic):
        ct = np.zeros((4, 64))
        for r in range(4):
            rho = ic * 16 + 4 * G4 + r
            ok = rho < C
            rr = np.where(ok, rho, 0)
            ct[r] = np.where(ok, (x16[S16, rr] - cst[d["c_xmean"] + rr]) / cst[d["c_xstd"] + rr], 0.0)
        return ct
    def slot_val(t4, sl):
        v = t4[sl & 3]
        return v[(LANES & 15) + 16 * (sl >> 2)]
    def scale(a):
        return (softplus(a) if d.get("scale_fn", 0) == 0 else 1 / (1 + np.exp(-(a + 2.0)))) + 1e-3
    for t in range(T - 1, -1, -1):
        tp = t * d["t16_stride"]
        c0 = np.zeros((NT, 4, 64))
        for ot in range(NT):
            for r in range(4):
                c0[ot, r] = packed[tp + d["o16_b0"] + (ot * 4 + G4) * 4 + r]
            for ic in range(d["nC16"]):
                mma16(packed, tp + d["o16_wc"], d["nC16"], ot, ic, ctx_tile(ic), c0[ot])
        act = np.zeros((3, NT, 4, 64))
        ut = np.zeros((4, 64))
        hdone = np.zeros((4, 64))
        if head_mfma:
            for r in range(4):
                hdone[r] = packed[tp + d["o16_bh"] + G4 * 4 + r]
        for p in range(1, D + 1):
            sl = int(cst[d["c_dslot"] + t * 16 + (p - 1)])
            pa = np.zeros(64); pm = np.zeros(64)
            fresh = None
            if p >= 2:
                lo, hi = d["g16_lo"][p - 1], d["g16_tile"][p - 1]     # tiles holding the group of degree p-1
                for ot in range(lo, hi + 1):
                    act[0, ot] = c0[ot]
                    mma16(packed, tp + d["o16_w0"], 1, ot, 0, ut, act[0, ot])
                for k in range(NB):
                    new = {}
                    for ot in range(lo, hi + 1):
                        b = np.zeros((4, 64))
                        for r in range(4):
                            b[r] = packed[tp + d[f"o16_bk{k}"] + (ot * 4 + G4) * 4 + r]
                        for it in range(hi + 1):
                            mma16(packed, tp + d[f"o16_wk{k}"], NT, ot, it, act[k, it], b)
                        new[ot] = np.tanh(b / TANH_PRESCALE)   # (weights and biases of the hidden blocks are packed pre-scaled)
                    for ot, v in new.items():
                        act[k + 1, ot] = v
                if head_mfma:
                    assert lo == hi, "the MFMA head exists for the aligned placement only"
                    fresh = hdone.copy()
                    mma16(packed, tp + d["o16_wh"], NT, 0, hi, act[NB, hi], fresh)
                    nxt = d["g16_tile"][p] if p < D else -1
                    if nxt != hi:
                        hdone = fresh
                hv = tp + d["o16_hv"] + sl * 128 + G4 * 32          # [slot][g4][tile][r][a|m]
                for tl in range(hi + 1):
                    for r in range(4):
                        pa += packed[hv + tl * 8 + 2 * r] * act[NB, tl, r]
                        pm += packed[hv + tl * 8 + 2 * r + 1] * act[NB, tl, r]
            def sum4(v):
                v = v + v[LANES ^ 16]
                return v + v[LANES ^ 32]
            if fresh is not None:   # rows 2 sl, 2 sl + 1 of the head tile: row group sl >> 1, registers 0,1 or 2,3
                src = (LANES & 15) + 16 * (sl >> 1)
                av = fresh[(sl & 1) * 2][src]
                mv = fresh[(sl & 1) * 2 + 1][src]
            else:
                av = packed[tp + d["o16_hvb"] + 2 * sl] + sum4(pa)
                mv = packed[tp + d["o16_hvb"] + 2 * sl + 1] + sum4(pm)
            wv = (slot_val(u, sl) - mv) / scale(av)
            for r in range(4):
                ut[r] = np.where((G4 == (sl >> 2)) & (r == (sl & 3)), wv, ut[r])
        u = ut
    th = np.zeros((16, D))
    for r in range(4):
        for g in range(4):
            p = 4 * g + r
            if p < D:
                td = int(cst[d["c_tdim"] + p])
                lanes = np.arange(16) + 16 * g
                th[:, td] = (u[r][lanes] - cst[d["c_pshift"] + p]) / cst[d["c_pscale"] + p]
    return th

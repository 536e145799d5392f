"""numpy model of the hand-derived rational-quadratic spline backward used by the HIP NSF training
kernel (synference_amd/csrc/sf_train_nsf.h).  Test infrastructure: checked against torch.autograd on
the oracle's rq_spline in tests/test_cpu_spline_backward.py."""
import numpy as np


def softplus(x):
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


def spline_fwd_bwd(q, v, Go, Gl, K, H, B=3.0, mw=1e-3, mh=1e-3, md=1e-3):
    """q [3K-1], scalar v; returns out, lad, dL/dv, dL/dq for L = Go*out + Gl*lad."""
    uw, uh, ud = q[:K], q[K:2 * K], q[2 * K:]
    inside = (v >= -B) and (v <= B)
    dq = np.zeros_like(q)
    if not inside:
        return v, 0.0, Go, dq

    def family(logits, msz):
        a = logits / np.sqrt(H)
        e = np.exp(a - a.max())
        p = e / e.sum()
        wn = msz + (1 - msz * K) * p
        cs = np.cumsum(wn)
        c = np.concatenate([[-B], 2 * B * cs[:-1] - B, [B]])
        return p, c

    pw, cw = family(uw, mw)
    ph, ch = family(uh, mh)
    idx = 0
    for k in range(K):
        if v >= cw[k]:
            idx = k
    x_k, w_k = cw[idx], cw[idx + 1] - cw[idx]
    y_k, h_k = ch[idx], ch[idx + 1] - ch[idx]
    const = np.log(np.exp(1 - md) - 1)
    udp = np.concatenate([[const], ud, [const]])
    der = md + softplus(udp)
    d_k, d_k1 = der[idx], der[idx + 1]
    s = h_k / w_k
    xi = (v - x_k) / w_k
    om = xi * (1 - xi)
    A = d_k + d_k1 - 2 * s
    N = s * xi * xi + d_k * om
    den = s + A * om
    out = y_k + h_k * N / den
    M = d_k1 * xi * xi + 2 * s * om + d_k * (1 - xi) ** 2
    dnum = s * s * M
    lad = np.log(dnum) - 2 * np.log(den)
    # ---- partials wrt z in {s, d_k, d_k1, xi}
    N_s, N_dk, N_dk1, N_xi = xi * xi, om, 0.0, 2 * s * xi + d_k * (1 - 2 * xi)
    D_s, D_dk, D_dk1, D_xi = 1 - 2 * om, om, om, A * (1 - 2 * xi)
    Q_s = 2 * s * M + s * s * 2 * om
    Q_dk = s * s * (1 - xi) ** 2
    Q_dk1 = s * s * xi * xi
    Q_xi = s * s * (2 * d_k1 * xi + 2 * s * (1 - 2 * xi) - 2 * d_k * (1 - xi))

    def Lz(Nz, Dz, Qz):
        return Go * h_k * (Nz * den - N * Dz) / (den * den) + Gl * (Qz / dnum - 2 * Dz / den)

    L_s, L_dk, L_dk1, L_xi = Lz(N_s, D_s, Q_s), Lz(N_dk, D_dk, Q_dk), Lz(N_dk1, D_dk1, Q_dk1), Lz(N_xi, D_xi, Q_xi)
    L_y = Go
    L_h = Go * N / den + L_s / w_k
    L_w = -L_s * s / w_k - L_xi * xi / w_k
    L_x = -L_xi / w_k
    L_v = L_xi / w_k

    def family_bwd(p, L_left, L_size, msz):
        """left knot c_idx and size c_{idx+1}-c_idx -> gradient wrt the raw logits of the family."""
        Lc0 = L_left - L_size      # wrt c_idx
        Lc1 = L_size               # wrt c_{idx+1}
        dwn = np.zeros(K)
        for i in range(K):
            g = 0.0
            if idx >= 1 and idx - 1 >= i:
                g += Lc0
            if idx + 1 <= K - 1 and idx >= i:
                g += Lc1
            dwn[i] = 2 * B * g
        dp = (1 - msz * K) * dwn
        da = p * (dp - (p * dp).sum())
        return da / np.sqrt(H)

    dq[:K] = family_bwd(pw, L_x, L_w, mw)
    dq[K:2 * K] = family_bwd(ph, L_y, L_h, mh)
    for j in range(1, K):
        g = (L_dk if j == idx else 0.0) + (L_dk1 if j == idx + 1 else 0.0)
        dq[2 * K + j - 1] = g * sigmoid(ud[j - 1])
    return out, lad, L_v, dq

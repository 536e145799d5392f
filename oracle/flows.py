"""CPU restatement (torch, fp64 or fp32) of the conditional flows on the hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py ("parity unpinned").

What is restated (all [UPSTREAM], reached from the reference at
src/synference/sbi_runner.py:5123-5146 and src/synference/custom_runner.py:320-326
through ``ili.utils.load_nde_sbi`` -> ``sbi.utils.posterior_nn``):

* sbi ``build_maf``: pointwise standardising affine, then T x
  [nflows MaskedAffineAutoregressiveTransform(hidden=H, num_blocks=2,
  use_residual_blocks=False, tanh), RandomPermutation]          
* sbi ``build_nsf``: standardising affine, then T x
  [PiecewiseRationalQuadraticCouplingTransform(alternating mask, ResidualNet
  conditioner with GLU context gates, K bins, linear tails, bound 3), LULinear]
                                                                
* nflows ``Flow.log_prob`` / ``Flow._sample`` with a StandardNormal base (B.5)

The trainable parameters are ONE flat vector in the "logical layout" defined by
``param_layout`` below; ``include/synference_hip.h`` documents the same layout for
``sf_flow_set_params``.  All functions are plain differentiable torch code so
``torch.autograd`` provides reference gradients for the HIP backward kernels.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# specification
# --------------------------------------------------------------------------------------
@dataclass
class FlowSpec:
    """Static description of one flow (hyper-parameters + non-trainable buffers).

    Defaults are the upstream defaults listed in SURVEY.md section 7 "Hard parts";
    every constant is overridable because upstream versions differ.
    """

    kind: str  # "maf" | "nsf" | "nsf_ar" (zuko / lampe autoregressive NSF) | "maf_ar" (zuko / lampe MAF)
    D: int  # theta dimension (flow inputs)
    C: int  # context width seen by the transforms
    H: int = 50  # hidden_features (sbi_runner.py:4402)
    T: int = 5  # num_transforms
    K: int = 10  # num_bins (sbi default 10; BASELINE cfg3 uses 8)
    NB: int = 2  # num_blocks
    tail_bound: float = 3.0
    min_bin_width: float = 1e-3
    min_bin_height: float = 1e-3
    min_derivative: float = 1e-3
    maf_eps: float = 1e-3
    lu_eps: float = 1e-3
    scale_fn: str = "softplus"  # nflows>=0.14; "sigmoid2" = sigmoid(a+2) (nflows<=0.13)
    hidden_bf16: bool = False  # emulate the HIP bf16 mode: hidden HxH operands rounded to bf16, wide accumulate
    ar_slope: float = 1e-3  # nsf_ar: zuko MonotonicRQSTransform(slope=...): soft clip of the spline logits
    theta_mean: Optional[np.ndarray] = None
    theta_std: Optional[np.ndarray] = None
    x_mean: Optional[np.ndarray] = None
    x_std: Optional[np.ndarray] = None
    perms: Optional[np.ndarray] = None  # [T, D] int64, MAF RandomPermutation buffers

    def __post_init__(self):
        assert self.kind in ("maf", "nsf", "nsf_ar", "maf_ar")
        if self.theta_mean is None:
            self.theta_mean = np.zeros(self.D)
        if self.theta_std is None:
            self.theta_std = np.ones(self.D)
        if self.x_mean is None:
            self.x_mean = np.zeros(self.C)
        if self.x_std is None:
            self.x_std = np.ones(self.C)
        if self.perms is None:
            self.perms = np.tile(np.arange(self.D), (self.T, 1))
        self.theta_mean = np.asarray(self.theta_mean, dtype=np.float64)
        self.theta_std = np.asarray(self.theta_std, dtype=np.float64)
        self.x_mean = np.asarray(self.x_mean, dtype=np.float64)
        self.x_std = np.asarray(self.x_std, dtype=np.float64)
        self.perms = np.asarray(self.perms, dtype=np.int64).reshape(self.T, self.D)

    # NSF coupling split for transform t (sbi build_nsf mask_in_layer +
    # nflows create_alternating_binary_mask): transform dims = mask>0.
    def nsf_split(self, t: int) -> Tuple[List[int], List[int]]:
        start = 0 if t % 2 == 0 else 1
        tr = list(range(start, self.D, 2))
        idn = [d for d in range(self.D) if d not in tr]
        return idn, tr

    @property
    def ar_np(self) -> int:
        """Parameters per dimension of the zuko-style autoregressive flows: 3K - 1 spline slots (nsf_ar), shift + scale (maf_ar)."""
        return 2 if self.kind == "maf_ar" else 3 * self.K - 1

    @property
    def has_lu(self) -> bool:
        return self.kind == "nsf" and self.D > 1

    @property
    def nsf_1d(self) -> bool:
        """[UPSTREAM] sbi build_nsf, `if x_numel == 1`: the coupling mask is [1] in EVERY transform (the single dimension is
        always transformed, nothing is left to condition on) and the spline parameters come from the context alone through
        ContextSplineMap; reached through the same load_nde_sbi call (ref: sbi_runner.py:5121-5146) for a one-parameter fit."""
        return self.kind == "nsf" and self.D == 1


def standardize_stats(theta: np.ndarray, x: np.ndarray) -> Dict[str, np.ndarray]:
    """z-score buffers as sbi builds them: unbiased std, clamped.

    theta: std < 1e-14 -> 1e-14 ; x: std < 1e-7 -> 1e-7.  Statistics are computed in
    float32 like upstream (tensors are float32 there: custom_runner.py:164-165).
    """
    t = torch.as_tensor(np.asarray(theta), dtype=torch.float32)
    xx = torch.as_tensor(np.asarray(x), dtype=torch.float32)
    t_mean, t_std = t.mean(0), t.std(0)
    t_std = torch.where(t_std < 1e-14, torch.full_like(t_std, 1e-14), t_std)
    x_mean, x_std = xx.mean(0), xx.std(0)
    x_std = torch.where(x_std < 1e-7, torch.full_like(x_std, 1e-7), x_std)
    return dict(
        theta_mean=t_mean.double().numpy(),
        theta_std=t_std.double().numpy(),
        x_mean=x_mean.double().numpy(),
        x_std=x_std.double().numpy(),
    )


# --------------------------------------------------------------------------------------
# logical parameter layout
# --------------------------------------------------------------------------------------
def param_layout(spec: FlowSpec) -> List[Tuple[str, Tuple[int, ...], int]]:
    """[(name, shape, offset)] of every trainable tensor in the flat vector.

    Order per transform t (row-major tensors, torch ``nn.Linear`` [out, in] weights):
      MAF: W0[H,D] b0[H] Wc[H,C] bc[H] {Wk[H,H] bk[H]}xNB Wf[2D,H] bf[2D]
      NSF: Win[H,d_id+C] bin[H] {Wg[H,C] bg[H] W1[H,H] b1[H] W2[H,H] b2[H]}xNB
           Wout[d_tr*(3K-1),H] bout[...]  then (D>1) LU: lower[D(D-1)/2]
           upper[D(D-1)/2] udiag[D] lubias[D]
      NSF with D = 1: csm.W0[H,C] b0[H] W1[H,H] b1[H] W2[3K-1,H] b2[3K-1]
      nsf_ar (zuko): ar.W0[H,D+C] b0[H] {ar.Wk[H,H] bk[H]} x (NB-1)  ar.W_NB[D(3K-1),H] b_NB[D(3K-1)]
      maf_ar (zuko): the same with 2 rows per dimension in the head ([shift, scale] of MonotonicAffineTransform)
    cfg1 MAF: 6460 per transform; cfg3 NSF: 18314 per transform.
    """
    out: List[Tuple[str, Tuple[int, ...], int]] = []
    off = 0

    def add(name, shape):
        nonlocal off
        out.append((name, tuple(shape), off))
        off += int(np.prod(shape))

    D, C, H = spec.D, spec.C, spec.H
    for t in range(spec.T):
        p = f"t{t}."
        if spec.kind == "maf":
            add(p + "W0", (H, D)); add(p + "b0", (H,))
            add(p + "Wc", (H, C)); add(p + "bc", (H,))
            for k in range(spec.NB):
                add(p + f"W{k + 1}", (H, H)); add(p + f"b{k + 1}", (H,))
            add(p + "Wf", (2 * D, H)); add(p + "bf", (2 * D,))
        elif spec.kind in ("nsf_ar", "maf_ar"):
            # zuko MaskedMLP hyper-network of one MaskedAutoregressiveTransform: NB hidden MaskedLinear layers of width H
            # (lampe / ltu-ili: two), then the head with D * ar_np rows -- per dimension [K widths, K heights, K - 1 derivatives]
            # (NSF) or [shift, scale] (MAF)
            add(p + "ar.W0", (H, D + C)); add(p + "ar.b0", (H,))
            for k in range(1, spec.NB):
                add(p + f"ar.W{k}", (H, H)); add(p + f"ar.b{k}", (H,))
            add(p + f"ar.W{spec.NB}", (D * spec.ar_np, H)); add(p + f"ar.b{spec.NB}", (D * spec.ar_np,))
        elif spec.nsf_1d:
            # sbi build_nsf with a scalar theta: ContextSplineMap(hidden_layers=1) -- Linear(C, H), ReLU, Linear(H, H), ReLU,
            # Linear(H, 3K - 1) on the embedded context alone; no LULinear
            add(p + "csm.W0", (H, C)); add(p + "csm.b0", (H,))
            add(p + "csm.W1", (H, H)); add(p + "csm.b1", (H,))
            add(p + "csm.W2", (3 * spec.K - 1, H)); add(p + "csm.b2", (3 * spec.K - 1,))
        else:
            idn, tr = spec.nsf_split(t)
            nout = len(tr) * (3 * spec.K - 1)
            add(p + "Win", (H, len(idn) + C)); add(p + "bin", (H,))
            for k in range(spec.NB):
                add(p + f"blk{k}.Wg", (H, C)); add(p + f"blk{k}.bg", (H,))
                add(p + f"blk{k}.W1", (H, H)); add(p + f"blk{k}.b1", (H,))
                add(p + f"blk{k}.W2", (H, H)); add(p + f"blk{k}.b2", (H,))
            add(p + "Wout", (nout, H)); add(p + "bout", (nout,))
            if spec.has_lu:
                nl = D * (D - 1) // 2
                add(p + "lu.lower", (nl,)); add(p + "lu.upper", (nl,))
                add(p + "lu.udiag", (D,)); add(p + "lu.bias", (D,))
    return out


def num_params(spec: FlowSpec) -> int:
    lay = param_layout(spec)
    name, shape, off = lay[-1]
    return off + int(np.prod(shape))


def views(spec: FlowSpec, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
    return {n: flat[o:o + int(np.prod(s))].view(*s) for n, s, o in param_layout(spec)}


def init_params(spec: FlowSpec, seed: int = 42) -> np.ndarray:
    """Seeded initialisation following the upstream initialisers.

    ``nn.Linear`` default = U(+-1/sqrt(fan_in)) for weight and bias (masks do not
    change fan_in); ResidualBlock's last linear U(+-1e-3); LULinear identity_init.
    The random stream is this oracle's own (numpy PCG64) -- upstream's torch stream
    is not reproducible here and is not part of the parity contract.
    """
    rng = np.random.default_rng(seed)
    flat = np.zeros(num_params(spec), dtype=np.float64)
    for name, shape, off in param_layout(spec):
        n = int(np.prod(shape))
        leaf = name.split(".")[-1]
        if name.endswith("lu.lower") or name.endswith("lu.upper") or name.endswith("lu.bias"):
            v = np.zeros(n)
        elif name.endswith("lu.udiag"):
            v = np.full(n, math.log(math.exp(1.0 - spec.lu_eps) - 1.0))
        elif leaf in ("W2", "b2") and ".blk" in name:
            v = rng.uniform(-1e-3, 1e-3, n)
        else:
            if leaf.startswith("W"):
                fan_in = shape[1]
                wname = name
            else:
                # bias: fan_in of its weight
                wname = name[: -len(leaf)] + "W" + leaf[1:]
                fan_in = [s for nn_, s, _ in param_layout(spec) if nn_ == wname][0][1]
            b = 1.0 / math.sqrt(fan_in)
            v = rng.uniform(-b, b, n)
        flat[off:off + n] = v
    return flat


def random_perms(spec_D: int, T: int, seed: int = 42) -> np.ndarray:
    rng = np.random.default_rng(seed + 7919)
    return np.stack([rng.permutation(spec_D) for _ in range(T)])


# --------------------------------------------------------------------------------------
# MADE degrees / masks  (nflows MaskedLinear._get_mask_and_degrees)
# --------------------------------------------------------------------------------------
def made_degrees(D: int, H: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    deg_in = np.arange(1, D + 1)
    max_, min_ = max(1, D - 1), min(1, D - 1)
    deg_h = np.arange(H) % max_ + min_
    deg_out = np.repeat(np.arange(1, D + 1), 2)  # [1,1,2,2,...]
    return deg_in, deg_h, deg_out


def made_masks(D: int, H: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    deg_in, deg_h, deg_out = made_degrees(D, H)
    M0 = (deg_h[:, None] >= deg_in[None, :]).astype(np.float64)  # [H, D]
    Mh = (deg_h[:, None] >= deg_h[None, :]).astype(np.float64)  # [H, H]
    Mf = (deg_out[:, None] > deg_h[None, :]).astype(np.float64)  # [2D, H]
    return M0, Mh, Mf


# --------------------------------------------------------------------------------------
# shared pieces
# --------------------------------------------------------------------------------------
def _t(a, like: torch.Tensor) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a), dtype=like.dtype)


def embed_context(spec: FlowSpec, x: torch.Tensor) -> torch.Tensor:
    """sbi ``Standardize`` in front of the (identity) embedding net."""
    return (x - _t(spec.x_mean, x)) / _t(spec.x_std, x)


def _scale_from_unconstrained(spec: FlowSpec, a: torch.Tensor) -> torch.Tensor:
    if spec.scale_fn == "softplus":
        return F.softplus(a) + spec.maf_eps
    return torch.sigmoid(a + 2.0) + spec.maf_eps


def _bf(spec: FlowSpec, t: torch.Tensor) -> torch.Tensor:
    """bf16 rounding (round to nearest even) of an MFMA operand when the bf16 mode is emulated."""
    return t.to(torch.bfloat16).to(t.dtype) if spec.hidden_bf16 else t


def _made(spec: FlowSpec, P: Dict[str, torch.Tensor], t: int, u: torch.Tensor,
          e: torch.Tensor, masks) -> Tuple[torch.Tensor, torch.Tensor]:
    """nflows MADE.forward (feed-forward blocks): returns (a, m) each [B, D]."""
    M0, Mh, Mf = masks
    p = f"t{t}."
    h = F.linear(u, P[p + "W0"] * M0, P[p + "b0"]) + F.linear(e, P[p + "Wc"], P[p + "bc"])
    for k in range(spec.NB):
        h = torch.tanh(F.linear(_bf(spec, h), _bf(spec, P[p + f"W{k + 1}"] * Mh), P[p + f"b{k + 1}"]))
    out = F.linear(h, P[p + "Wf"] * Mf, P[p + "bf"]).view(-1, spec.D, 2)
    return out[..., 0], out[..., 1]


def _resnet(spec: FlowSpec, P: Dict[str, torch.Tensor], t: int, u_id: torch.Tensor,
            e: torch.Tensor) -> torch.Tensor:
    """nflows ResidualNet.forward with GLU context gates."""
    p = f"t{t}."
    h = F.linear(torch.cat([u_id, e], dim=1), P[p + "Win"], P[p + "bin"])
    for k in range(spec.NB):
        b = p + f"blk{k}."
        tt = F.linear(_bf(spec, F.relu(h)), _bf(spec, P[b + "W1"]), P[b + "b1"])
        tt = F.linear(_bf(spec, F.relu(tt)), _bf(spec, P[b + "W2"]), P[b + "b2"])
        tt = tt * torch.sigmoid(F.linear(e, P[b + "Wg"], P[b + "bg"]))
        h = h + tt
    return F.linear(h, P[p + "Wout"], P[p + "bout"])


def _context_spline_map(spec: FlowSpec, P: Dict[str, torch.Tensor], t: int, e: torch.Tensor) -> torch.Tensor:
    """[UPSTREAM] sbi ContextSplineMap.__call__: ``spline_predictor(context)``, the inputs are ignored."""
    p = f"t{t}.csm."
    h = F.relu(F.linear(e, P[p + "W0"], P[p + "b0"]))
    h = F.relu(F.linear(h, P[p + "W1"], P[p + "b1"]))
    return F.linear(h, P[p + "W2"], P[p + "b2"])


# --------------------------------------------------------------------------------------
# zuko / lampe autoregressive NSF  (ref: sbi_runner.py:5123-5125 `ili.utils.load_nde_lampe`;
# examples/sbi/scripts/basic_model.py:31-41 trains a three-member ensemble of them)
# --------------------------------------------------------------------------------------
# [UPSTREAM, restated from the published zuko sources (zuko.flows.NSF -> MAF -> MaskedAutoregressiveTransform -> MaskedMLP,
#  zuko.transforms.MonotonicRQSTransform); parity unpinned like every other flow here]
#   * transforms t = 0 .. T-1 alternate the ordering: order_t = arange(D) for even t, reversed for odd t (randperm=False);
#   * hyper-network input [theta ; context], adjacency A[o, i] = order[o // (3K-1)] > in_order[i], in_order = [order, -1 x C]:
#     the parameters of dimension d see the dimensions ordered before it and the whole context;
#   * MaskedMLP: the UNIQUE rows of A (one per order value r, sorted: index r) are the unit "types"; hidden unit h of every
#     hidden layer has type h mod D; first-layer mask = A-row of the type; later masks: type(in) <= type(out) (row inclusion);
#     head row of a dimension with order value r: hidden types <= r.  ReLU between layers;
#   * MonotonicRQSTransform(widths, heights, derivatives, bound = 5, slope = 1e-3): logits soft-clipped
#     w / (1 + |2 w / log slope|) (derivatives: d / (1 + |d / log slope|)), softmax -> knots on [-bound, bound] (no minimum bin
#     size), knot derivatives exp(.) with 1 at both ends, identity outside the bound; bin = searchsorted(knots, x) - 1.
def ar_order(spec: FlowSpec, t: int) -> np.ndarray:
    o = np.arange(spec.D)
    return o if t % 2 == 0 else o[::-1].copy()


def ar_masks(spec: FlowSpec, t: int) -> List[np.ndarray]:
    """Boolean masks [first hidden (H, D+C), later hidden (H, H) x (NB-1), head (D(3K-1), H)] of transform t."""
    D, C, H, NP = spec.D, spec.C, spec.H, spec.ar_np
    order = ar_order(spec, t)
    in_order = np.concatenate([order, np.full(C, -1)])
    typ = np.arange(H) % D                                   # type (= order value) of hidden unit h
    masks = [typ[:, None] > in_order[None, :]]               # A-row of the type
    for _ in range(1, spec.NB):
        masks.append(typ[:, None] >= typ[None, :])           # row inclusion: type(in) <= type(out)
    out_type = np.repeat(order, NP)
    masks.append(out_type[:, None] >= typ[None, :])
    return masks


def _ar_hyper(spec: FlowSpec, P: Dict[str, torch.Tensor], t: int, u: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    p = f"t{t}.ar."
    M = [_t(m.astype(np.float64), u) for m in ar_masks(spec, t)]
    h = torch.cat([u, e], dim=1)
    for k in range(spec.NB + 1):
        h = F.linear(h, P[p + f"W{k}"] * M[k], P[p + f"b{k}"])
        if k < spec.NB:
            h = F.relu(h)
    return h.view(-1, spec.D, spec.ar_np)


def ar_affine(spec: FlowSpec, v: torch.Tensor, q: torch.Tensor, inverse: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """[UPSTREAM, restated from the published zuko sources] zuko.transforms.MonotonicAffineTransform(shift, scale, slope) -- the
    univariate map of zuko.flows.MAF (`backend="lampe"`, model "maf": ref sbi_runner.py:5123-5125): the scale logit is soft-clipped,
    log_scale = s / (1 + |s / log slope|), y = x exp(log_scale) + shift, log|dy/dx| = log_scale; v [B, d], q [B, d, 2] = [shift, s]."""
    shift, sraw = q[..., 0], q[..., 1]
    ls = sraw / (1 + torch.abs(sraw / math.log(spec.ar_slope)))
    if inverse:
        return (v - shift) * torch.exp(-ls), -ls
    return v * torch.exp(ls) + shift, ls


def _ar_knots(spec: FlowSpec, q: torch.Tensor):
    """(horizontal, vertical, derivatives) of zuko's MonotonicRQSTransform from the raw head outputs q[..., 3K-1]."""
    K, B = spec.K, spec.tail_bound
    ls = math.log(spec.ar_slope)
    w, hh, d = q[..., :K], q[..., K:2 * K], q[..., 2 * K:]
    w = w / (1 + torch.abs(2 * w / ls))
    hh = hh / (1 + torch.abs(2 * hh / ls))
    d = d / (1 + torch.abs(d / ls))
    w = F.pad(F.softmax(w, dim=-1), (1, 0), value=0.0)
    hh = F.pad(F.softmax(hh, dim=-1), (1, 0), value=0.0)
    d = F.pad(d, (1, 1), value=0.0)
    return B * (2 * torch.cumsum(w, dim=-1) - 1), B * (2 * torch.cumsum(hh, dim=-1) - 1), torch.exp(d)


def ar_spline(spec: FlowSpec, v: torch.Tensor, q: torch.Tensor, inverse: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """zuko MonotonicRQSTransform._call / ._inverse with log|det|; v [B, d], q [B, d, 3K-1]."""
    K = spec.K
    hor, ver, der = _ar_knots(spec, q)
    seq = (ver if inverse else hor).contiguous()
    k = torch.searchsorted(seq, v[..., None].contiguous()).squeeze(-1) - 1
    mask = (k >= 0) & (k < K)
    k = k % K
    g = lambda a, kk: a.gather(-1, kk[..., None]).squeeze(-1)
    x0, x1, y0, y1, d0, d1 = g(hor, k), g(hor, k + 1), g(ver, k), g(ver, k + 1), g(der, k), g(der, k + 1)
    s = (y1 - y0) / (x1 - x0)
    if not inverse:
        z = mask * (v - x0) / (x1 - x0)
        y = y0 + (y1 - y0) * (s * z ** 2 + d0 * z * (1 - z)) / (s + (d0 + d1 - 2 * s) * z * (1 - z))
        out = torch.where(mask, y, v)
    else:
        y_ = mask * (v - y0)
        a = (y1 - y0) * (s - d0) + y_ * (d0 + d1 - 2 * s)
        b = (y1 - y0) * d0 - y_ * (d0 + d1 - 2 * s)
        c = -s * y_
        z = 2 * c / (-b - torch.sqrt(b ** 2 - 4 * a * c))
        out = torch.where(mask, x0 + z * (x1 - x0), v)
    jac = s ** 2 * (2 * s * z * (1 - z) + d0 * (1 - z) ** 2 + d1 * z ** 2) / (s + (d0 + d1 - 2 * s) * z * (1 - z)) ** 2
    lad = mask * torch.log(jac)
    return out, (-lad if inverse else lad)


def _knots(spec: FlowSpec, logits: torch.Tensor, min_size: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """softmax -> min size -> cumsum -> [-B, B] knots; returns (knots[...,K+1], sizes[...,K])."""
    K, B = spec.K, spec.tail_bound
    w = F.softmax(logits, dim=-1)
    w = min_size + (1.0 - min_size * K) * w
    cw = torch.cumsum(w, dim=-1)
    cw = F.pad(cw, (1, 0), value=0.0)
    cw = 2.0 * B * cw - B
    cw = torch.cat([torch.full_like(cw[..., :1], -B), cw[..., 1:-1], torch.full_like(cw[..., :1], B)], dim=-1)
    return cw, cw[..., 1:] - cw[..., :-1]


def rq_spline(spec: FlowSpec, v: torch.Tensor, q: torch.Tensor, inverse: bool
              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """nflows unconstrained_rational_quadratic_spline, tails='linear' (SURVEY.md B.4).

    v: [B, d] inputs ; q: [B, d, 3K-1] raw conditioner outputs.  Returns (out, logabsdet[B, d]).
    """
    K, B = spec.K, spec.tail_bound
    uw = q[..., :K] / math.sqrt(spec.H)
    uh = q[..., K:2 * K] / math.sqrt(spec.H)
    ud = q[..., 2 * K:]
    const = math.log(math.exp(1.0 - spec.min_derivative) - 1.0)
    ud = torch.cat([torch.full_like(ud[..., :1], const), ud, torch.full_like(ud[..., :1], const)], dim=-1)
    cw, w = _knots(spec, uw, spec.min_bin_width)
    ch, hh = _knots(spec, uh, spec.min_bin_height)
    der = spec.min_derivative + F.softplus(ud)

    inside = (v >= -B) & (v <= B)
    vc = torch.clamp(v, -B, B)
    loc = ch if inverse else cw
    loc_s = torch.cat([loc[..., :-1], loc[..., -1:] + 1e-6], dim=-1)
    idx = (torch.sum(vc[..., None] >= loc_s, dim=-1) - 1).clamp(0, K - 1)[..., None]

    g = lambda a: a.gather(-1, idx)[..., 0]
    x_k, w_k, y_k, h_k = g(cw), g(w), g(ch), g(hh)
    delta = hh / w
    s_k, d_k, d_k1 = g(delta), g(der), g(der[..., 1:])

    if inverse:
        dy = vc - y_k
        tmp = dy * (d_k + d_k1 - 2 * s_k)
        a = tmp + h_k * (s_k - d_k)
        b = h_k * d_k - tmp
        c = -s_k * dy
        disc = b * b - 4 * a * c
        root = (2 * c) / (-b - torch.sqrt(disc))
        out = root * w_k + x_k
        xi = root
    else:
        xi = (vc - x_k) / w_k
        om = xi * (1 - xi)
        num = h_k * (s_k * xi * xi + d_k * om)
        den = s_k + (d_k + d_k1 - 2 * s_k) * om
        out = y_k + num / den
    om = xi * (1 - xi)
    den = s_k + (d_k + d_k1 - 2 * s_k) * om
    dnum = s_k * s_k * (d_k1 * xi * xi + 2 * s_k * om + d_k * (1 - xi) * (1 - xi))
    lad = torch.log(dnum) - 2 * torch.log(den)
    if inverse:
        lad = -lad
    out = torch.where(inside, out, v)
    lad = torch.where(inside, lad, torch.zeros_like(lad))
    return out, lad


def _lu_mats(spec: FlowSpec, P: Dict[str, torch.Tensor], t: int):
    D = spec.D
    p = f"t{t}.lu."
    li = np.tril_indices(D, -1)
    ui = np.triu_indices(D, 1)
    L = torch.eye(D, dtype=P[p + "lower"].dtype)
    L = L.index_put((torch.as_tensor(li[0]), torch.as_tensor(li[1])), P[p + "lower"])
    diag = F.softplus(P[p + "udiag"]) + spec.lu_eps
    U = torch.diag(diag)
    U = U.index_put((torch.as_tensor(ui[0]), torch.as_tensor(ui[1])), P[p + "upper"])
    return L, U, diag


# --------------------------------------------------------------------------------------
# density direction
# --------------------------------------------------------------------------------------
def forward_transform(spec: FlowSpec, flat: torch.Tensor, theta: torch.Tensor, x: torch.Tensor
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """theta -> z with total log|det J| (nflows CompositeTransform.forward)."""
    P = views(spec, flat)
    e = embed_context(spec, x)
    scale = 1.0 / _t(spec.theta_std, theta)
    shift = -_t(spec.theta_mean, theta) / _t(spec.theta_std, theta)
    u = theta * scale + shift
    logdet = torch.log(torch.abs(scale)).sum().expand(theta.shape[0]).clone()
    if spec.kind == "maf":
        masks = tuple(_t(m, theta) for m in made_masks(spec.D, spec.H))
        for t in range(spec.T):
            a, m = _made(spec, P, t, u, e, masks)
            s = _scale_from_unconstrained(spec, a)
            u = s * u + m
            logdet = logdet + torch.log(s).sum(-1)
            u = u[:, torch.as_tensor(spec.perms[t])]
    elif spec.kind in ("nsf_ar", "maf_ar"):
        uni = ar_spline if spec.kind == "nsf_ar" else ar_affine
        for t in range(spec.T):
            q = _ar_hyper(spec, P, t, u, e)
            u, lad = uni(spec, u, q, inverse=False)
            logdet = logdet + lad.sum(-1)
    else:
        for t in range(spec.T):
            if spec.nsf_1d:
                q = _context_spline_map(spec, P, t, e).view(-1, 1, 3 * spec.K - 1)
                u, lad = rq_spline(spec, u, q, inverse=False)
                logdet = logdet + lad.sum(-1)
                continue
            idn, tr = spec.nsf_split(t)
            q = _resnet(spec, P, t, u[:, idn], e).view(-1, len(tr), 3 * spec.K - 1)
            v, lad = rq_spline(spec, u[:, tr], q, inverse=False)
            u = u.clone()
            u[:, tr] = v
            logdet = logdet + lad.sum(-1)
            if spec.has_lu:
                L, U, diag = _lu_mats(spec, P, t)
                u = F.linear(F.linear(u, U), L, P[f"t{t}.lu.bias"])
                logdet = logdet + torch.log(diag).sum()
    return u, logdet


def log_prob(spec: FlowSpec, flat: torch.Tensor, theta: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """nflows Flow.log_prob: log N(z;0,I) + sum logdet  (SURVEY.md B.5)."""
    z, logdet = forward_transform(spec, flat, theta, x)
    return -0.5 * (z * z).sum(-1) - 0.5 * spec.D * LOG_2PI + logdet


# --------------------------------------------------------------------------------------
# sampling direction
# --------------------------------------------------------------------------------------
def inverse_transform(spec: FlowSpec, flat: torch.Tensor, z: torch.Tensor, x: torch.Tensor
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """z -> theta; returns (theta, log|det J_inverse|)  (nflows CompositeTransform.inverse).

    MAF: D full MADE passes per transform exactly as AutoregressiveTransform.inverse.
    NSF: LU inverse by two triangular solves, then the inverse spline.
    """
    P = views(spec, flat)
    e = embed_context(spec, x)
    u = z
    logdet = torch.zeros(z.shape[0], dtype=z.dtype)
    if spec.kind == "maf":
        masks = tuple(_t(m, z) for m in made_masks(spec.D, spec.H))
        for t in reversed(range(spec.T)):
            inv = np.argsort(spec.perms[t])
            v = u[:, torch.as_tensor(inv)]
            w = torch.zeros_like(v)
            for _ in range(spec.D):
                a, m = _made(spec, P, t, w, e, masks)
                s = _scale_from_unconstrained(spec, a)
                w = (v - m) / s
            logdet = logdet - torch.log(s).sum(-1)
            u = w
    elif spec.kind in ("nsf_ar", "maf_ar"):
        uni = ar_spline if spec.kind == "nsf_ar" else ar_affine
        for t in reversed(range(spec.T)):
            # zuko AutoregressiveTransform._inverse: `passes` = D sweeps of the hyper-network, each inverting every dimension
            # with the parameters of the current iterate; after sweep j the dimensions of order < j are exact
            v = u
            w = torch.zeros_like(v)
            for _ in range(spec.D):
                q = _ar_hyper(spec, P, t, w, e)
                w, lad = uni(spec, v, q, inverse=True)
            logdet = logdet + lad.sum(-1)
            u = w
    else:
        for t in reversed(range(spec.T)):
            if spec.nsf_1d:
                q = _context_spline_map(spec, P, t, e).view(-1, 1, 3 * spec.K - 1)
                u, lad = rq_spline(spec, u, q, inverse=True)
                logdet = logdet + lad.sum(-1)
                continue
            idn, tr = spec.nsf_split(t)
            if spec.has_lu:
                L, U, diag = _lu_mats(spec, P, t)
                y = (u - P[f"t{t}.lu.bias"]).t()
                y = torch.linalg.solve_triangular(L, y, upper=False, unitriangular=True)
                y = torch.linalg.solve_triangular(U, y, upper=True)
                u = y.t()
                logdet = logdet - torch.log(diag).sum()
            q = _resnet(spec, P, t, u[:, idn], e).view(-1, len(tr), 3 * spec.K - 1)
            v, lad = rq_spline(spec, u[:, tr], q, inverse=True)
            u = u.clone()
            u[:, tr] = v
            logdet = logdet + lad.sum(-1)
    scale = 1.0 / _t(spec.theta_std, z)
    shift = -_t(spec.theta_mean, z) / _t(spec.theta_std, z)
    theta = (u - shift) / scale
    logdet = logdet - torch.log(torch.abs(scale)).sum()
    return theta, logdet

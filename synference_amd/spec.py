"""Flow specification, logical parameter layout and initialisation (host side).

Mirrors what ``sbi.utils.posterior_nn(model, hidden_features, num_transforms, num_bins,
z_score_theta="independent", z_score_x="independent")`` decides when the reference calls
``estimator_builder(batch_x=x_train, batch_theta=theta_train)``
(ref: src/synference/custom_runner.py:320-326; src/synference/sbi_runner.py:5123-5146):
shapes from the batch, z-score buffers from its mean / unbiased std, a fixed random
permutation per MAF block, and torch-default weight initialisation.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

KIND_ID = {"maf": 0, "nsf": 1, "nsf_ar": 2, "maf_ar": 3}


@dataclass
class FlowSpec:
    kind: str               # "maf" | "nsf" | "nsf_ar" / "maf_ar" (the autoregressive NSF / MAF of the lampe / zuko backend)
    D: int                  # theta dimension
    C: int                  # context width seen by the transforms
    H: int = 50             # hidden_features (ref default: sbi_runner.py:4402)
    T: int = 5              # num_transforms
    K: int = 10             # num_bins (sbi default; only NSF)
    NB: int = 2             # num_blocks
    tail_bound: float = 3.0
    min_bin_width: float = 1e-3
    min_bin_height: float = 1e-3
    min_derivative: float = 1e-3
    maf_eps: float = 1e-3
    lu_eps: float = 1e-3
    scale_fn: str = "softplus"   # "sigmoid2" = sigmoid(a+2) of nflows <= 0.13
    hidden_bf16: bool = False    # bf16 MFMA operands for the hidden HxH layers of the inference kernels
    ar_slope: float = 1e-3       # nsf_ar: zuko MonotonicRQSTransform(slope=): soft clip of the spline logits
    theta_mean: Optional[np.ndarray] = None
    theta_std: Optional[np.ndarray] = None
    x_mean: Optional[np.ndarray] = None
    x_std: Optional[np.ndarray] = None
    perms: Optional[np.ndarray] = None   # [T, D] MAF RandomPermutation buffers

    def __post_init__(self):
        if self.kind not in KIND_ID:
            raise ValueError(
                f"model '{self.kind}' is not built by the HIP backend: only 'maf', 'nsf', 'nsf_ar' and 'maf_ar' "
                "(NPE, direct sampling) are on the accelerated path")
        f = lambda a, n, fill: (np.full(n, fill, np.float32) if a is None
                                else np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(n)))
        self.theta_mean = f(self.theta_mean, self.D, 0.0)
        self.theta_std = f(self.theta_std, self.D, 1.0)
        self.x_mean = f(self.x_mean, self.C, 0.0)
        self.x_std = f(self.x_std, self.C, 1.0)
        if self.perms is None:
            self.perms = np.tile(np.arange(self.D, dtype=np.int32), (self.T, 1))
        self.perms = np.ascontiguousarray(np.asarray(self.perms, dtype=np.int32).reshape(self.T, self.D))

    def nsf_split(self, t: int) -> Tuple[List[int], List[int]]:
        start = 0 if t % 2 == 0 else 1
        tr = list(range(start, self.D, 2))
        return [d for d in range(self.D) if d not in tr], tr

    @property
    def ar_np(self) -> int:
        """Head rows per dimension of the zuko-style flows: 3K - 1 spline slots (nsf_ar), [shift, scale] (maf_ar)."""
        return 2 if self.kind == "maf_ar" else 3 * self.K - 1

    @property
    def has_lu(self) -> bool:
        return self.kind == "nsf" and self.D > 1

    @property
    def nsf_1d(self) -> bool:
        """One-parameter NSF: sbi's build_nsf transforms the single dimension in every block and takes the spline
        parameters from a context-only MLP (ContextSplineMap: Linear, ReLU, Linear, ReLU, Linear); no LULinear."""
        return self.kind == "nsf" and self.D == 1

    def to_dict(self) -> dict:
        d = {k: getattr(self, k) for k in ("kind", "D", "C", "H", "T", "K", "NB", "tail_bound", "min_bin_width",
                                           "min_bin_height", "min_derivative", "maf_eps", "lu_eps", "scale_fn", "hidden_bf16", "ar_slope")}
        for k in ("theta_mean", "theta_std", "x_mean", "x_std", "perms"):
            d[k] = getattr(self, k).tolist()
        return d

    @classmethod
    def from_dict(cls, d: dict) -> "FlowSpec":
        return cls(**d)


def zscore_stats(theta, x) -> Dict[str, np.ndarray]:
    """sbi ``standardizing_transform`` / ``standardizing_net`` statistics ([UPSTREAM]):
    mean and unbiased std over the rows handed to the builder, std clamped at 1e-14 (theta) / 1e-7 (x)."""
    t = torch.as_tensor(np.asarray(theta), dtype=torch.float32)
    xx = torch.as_tensor(np.asarray(x), dtype=torch.float32)
    t_std = t.std(0).clamp_min(1e-14) if t.shape[0] > 1 else torch.ones(t.shape[1])
    x_std = xx.std(0).clamp_min(1e-7) if xx.shape[0] > 1 else torch.ones(xx.shape[1])
    return dict(theta_mean=t.mean(0).numpy(), theta_std=t_std.numpy(),
                x_mean=xx.mean(0).numpy(), x_std=x_std.numpy())


def param_layout(spec: FlowSpec) -> List[Tuple[str, Tuple[int, ...], int]]:
    """[(name, shape, offset)] of the flat parameter vector (documented in include/synference_hip.h)."""
    out, off = [], 0

    def add(name, shape):
        nonlocal off
        out.append((name, tuple(shape), off))
        off += int(np.prod(shape))

    D, Cc, H = spec.D, spec.C, spec.H
    for t in range(spec.T):
        p = f"t{t}."
        if spec.kind == "maf":
            add(p + "W0", (H, D)); add(p + "b0", (H,)); add(p + "Wc", (H, Cc)); add(p + "bc", (H,))
            for k in range(spec.NB):
                add(p + f"W{k + 1}", (H, H)); add(p + f"b{k + 1}", (H,))
            add(p + "Wf", (2 * D, H)); add(p + "bf", (2 * D,))
        elif spec.kind in ("nsf_ar", "maf_ar"):   # zuko MaskedMLP hyper-network: [theta ; context] -> H x NB -> D (3K - 1) | D x 2
            add(p + "ar.W0", (H, D + Cc)); add(p + "ar.b0", (H,))
            for k in range(1, spec.NB):
                add(p + f"ar.W{k}", (H, H)); add(p + f"ar.b{k}", (H,))
            add(p + f"ar.W{spec.NB}", (D * spec.ar_np, H)); add(p + f"ar.b{spec.NB}", (D * spec.ar_np,))
        elif spec.nsf_1d:
            add(p + "csm.W0", (H, Cc)); add(p + "csm.b0", (H,)); add(p + "csm.W1", (H, H)); add(p + "csm.b1", (H,))
            add(p + "csm.W2", (3 * spec.K - 1, H)); add(p + "csm.b2", (3 * spec.K - 1,))
        else:
            idn, tr = spec.nsf_split(t)
            nout = len(tr) * (3 * spec.K - 1)
            add(p + "Win", (H, len(idn) + Cc)); add(p + "bin", (H,))
            for k in range(spec.NB):
                add(p + f"blk{k}.Wg", (H, Cc)); add(p + f"blk{k}.bg", (H,))
                add(p + f"blk{k}.W1", (H, H)); add(p + f"blk{k}.b1", (H,))
                add(p + f"blk{k}.W2", (H, H)); add(p + f"blk{k}.b2", (H,))
            add(p + "Wout", (nout, H)); add(p + "bout", (nout,))
            if spec.has_lu:
                nl = D * (D - 1) // 2
                add(p + "lu.lower", (nl,)); add(p + "lu.upper", (nl,))
                add(p + "lu.udiag", (D,)); add(p + "lu.bias", (D,))
    return out


def num_params(spec: FlowSpec) -> int:
    name, shape, off = param_layout(spec)[-1]
    return off + int(np.prod(shape))


def init_params(spec: FlowSpec, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """torch ``nn.Linear`` default init (U(+-1/sqrt(fan_in)) weight and bias), ResidualBlock's last
    linear U(+-1e-3), LULinear identity_init  ([UPSTREAM])."""
    lay = param_layout(spec)
    shapes = {n: s for n, s, _ in lay}
    flat = torch.zeros(num_params(spec), dtype=torch.float32)
    for name, shape, off in lay:
        n = int(np.prod(shape))
        leaf = name.split(".")[-1]
        if name.endswith(("lu.lower", "lu.upper", "lu.bias")):
            continue
        if name.endswith("lu.udiag"):
            flat[off:off + n] = math.log(math.exp(1.0 - spec.lu_eps) - 1.0)
            continue
        if ".blk" in name and leaf in ("W2", "b2"):
            bound = 1e-3
        else:
            wname = name if leaf.startswith("W") else name[: -len(leaf)] + "W" + leaf[1:]
            bound = 1.0 / math.sqrt(shapes[wname][1])
        flat[off:off + n] = (torch.rand(n, generator=generator) * 2.0 - 1.0) * bound
    return flat


def random_perms(D: int, T: int, generator: Optional[torch.Generator] = None) -> np.ndarray:
    """nflows RandomPermutation: one ``torch.randperm(D)`` per block, fixed at construction."""
    return np.stack([torch.randperm(D, generator=generator).numpy() for _ in range(T)]).astype(np.int32)


def state_dict_views(spec: FlowSpec, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Named views into the flat vector (for checkpoints / interoperability)."""
    return {n: flat[o:o + int(np.prod(s))].view(*s) for n, s, o in param_layout(spec)}

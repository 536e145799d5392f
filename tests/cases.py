"""Shared seeded test cases: (oracle spec, product spec, flat params, theta, x)."""
from __future__ import annotations

import numpy as np
import torch

from oracle import flows as OF
from synference_amd.spec import FlowSpec

# name -> (kind, D, C, H, T, K)
CASES = {
    "maf_cfg1": ("maf", 5, 10, 50, 5, 10),     # BASELINE cfg1/cfg2
    "nsf_cfg3": ("nsf", 8, 20, 50, 5, 8),      # BASELINE cfg3
    "maf_small": ("maf", 2, 3, 16, 2, 10),
    "maf_wide": ("maf", 12, 40, 69, 3, 10),    # HT=3 path
    "nsf_odd": ("nsf", 5, 10, 30, 3, 10),      # odd D: d_tr differs by parity; HT=1
    "nsf_k16": ("nsf", 3, 7, 64, 2, 16),       # PT=3 path
    "maf_d1": ("maf", 1, 4, 8, 2, 10),         # one parameter: no autoregressive inputs at all
    "maf_nb3": ("maf", 4, 6, 24, 2, 10, dict(NB=3)),            # num_blocks=3 -> full-pass inverse fallback
    "nsf_nb1": ("nsf", 4, 6, 24, 2, 5, dict(NB=1)),             # num_blocks=1, K=5
    "maf_sig2": ("maf", 3, 5, 20, 2, 10, dict(scale_fn="sigmoid2")),  # nflows<=0.13 scale parametrisation
    # 16-row sampler with degree groups that straddle tiles (contiguous packing, sf_pass16_span)
    "maf_span6": ("maf", 6, 10, 50, 3, 10),    # 5 groups of 10 in 4 tiles
    "maf_span_h64": ("maf", 8, 12, 64, 2, 10), # the example CLI width: 7 groups of 9-10
    "maf_d2_span": ("maf", 2, 4, 40, 2, 10),   # one group of 40 units over 3 tiles
    "maf_d4": ("maf", 4, 6, 40, 3, 10),        # 3 groups of 13-14, one per tile: the unrolled sampler kernel with DD = 4
    "maf_d3": ("maf", 3, 5, 26, 3, 10),        # 2 groups of 13, one per tile: DD = 3
    # the reference's example CLI trains 6 transforms (examples/sbi/scripts/train_model.py:56-57): the TS = 6 / TS = 8 instantiations
    # of the cooperative MAF training kernel (the a1 / a2 stash of transforms beyond five is partly in scratch)
    # the reference's example CLI shape itself (hidden_features=64, num_transforms=6: examples/sbi/scripts/train_model.py:56-57) on a
    # 7-parameter / 16-filter problem: SPAN placement (six degree groups of 10-11 units over four tiles) and T = 6 -- the eight-slot
    # instantiation of the cooperative training kernel with the deeper stash
    "maf_cli": ("maf", 7, 16, 64, 6, 10),
    "maf_t6": ("maf", 5, 10, 50, 6, 10),
    "maf_t8": ("maf", 4, 6, 40, 8, 10),
    "maf_nb1": ("maf", 4, 6, 48, 2, 10, dict(NB=1)),   # one hidden block, 3 full tiles: the NB = 1 instantiations of the 16-row sampler
    # cooperative NSF training kernel (sf_nsfc.hip) beyond cfg3: 8 spline-head tiles (K = 10: sbi's default), three hidden
    # tiles with a partial last one, 6 parameters; and the smallest coupling (one identity dimension, one input tile)
    "nsf_k10": ("nsf", 6, 12, 40, 2, 10),
    "nsf_d2": ("nsf", 2, 3, 20, 3, 8),
    # one parameter: sbi builds a ContextSplineMap flow (context-only MLP -> spline parameters, no LULinear): sf_nsf1.hip
    "nsf_d1": ("nsf", 1, 5, 24, 3, 8),
    # the width of the reference's production NSF (examples/sbi/configs/best_params.yaml: 69 hidden, K = 10): five hidden tiles
    # of 16 -> the five-wave form of the cooperative training kernel
    "nsf_h69": ("nsf", 8, 20, 69, 3, 10),
    # the autoregressive NSF of the reference's lampe backend (zuko.flows.NSF: 8 bins, bound 5; sf_nsfar.hip): the cfg1 parameter
    # space, a small ragged one (H not a multiple of D, K = 5), one parameter, and a wide one (8 parameters, 64 hidden)
    "nsfar_cfg1": ("nsf_ar", 5, 10, 50, 5, 8, dict(tail_bound=5.0)),
    "nsfar_small": ("nsf_ar", 3, 4, 17, 2, 5, dict(tail_bound=5.0)),
    "nsfar_d1": ("nsf_ar", 1, 6, 16, 3, 8, dict(tail_bound=5.0)),
    "nsfar_wide": ("nsf_ar", 8, 20, 64, 3, 8, dict(tail_bound=5.0, ar_slope=1e-2)),
    # sixteen units per type (all four k-steps of the 16-candidate sampler's hidden blocks: sf_nsfar16.hip, KS = 4) and two input tiles
    # (D + C = 21), two units per type at D = 8 (KS = 2, half-empty tiles)
    "nsfar_k4": ("nsf_ar", 4, 17, 64, 2, 6, dict(tail_bound=5.0)),
    "nsfar_thin": ("nsf_ar", 8, 5, 16, 3, 4, dict(tail_bound=5.0)),
    # two tiles per type on the register-tile sampler, three k-steps (D = 5: two workgroups per CU) / wider than the sampler takes
    # (33 units per type: the 64-sample kernel)
    "nsfar_two": ("nsf_ar", 5, 9, 112, 2, 8, dict(tail_bound=5.0)),
    "nsfar_33": ("nsf_ar", 3, 6, 99, 2, 6, dict(tail_bound=5.0)),
    # the widest member of the reference's own lampe example (examples/sbi/scripts/basic_model.py:31-41: hidden_features 180): seven
    # types of 25-26 units padded to 32 rows each; fits since the training sweep runs on two hidden buffers
    "nsfar_h180": ("nsf_ar", 7, 12, 180, 2, 8, dict(tail_bound=5.0)),
    # zuko.flows.MAF -- `backend="lampe"`, model "maf" (ref: sbi_runner.py:5123-5125): the same masked hyper-network, affine univariate
    "mafar_cfg1": ("maf_ar", 5, 10, 50, 5, 8, dict(tail_bound=5.0)),
    "mafar_small": ("maf_ar", 3, 4, 17, 2, 8, dict(tail_bound=5.0, ar_slope=1e-2)),
}


def make_case(name: str, seed: int = 0, B: int = 200, spread: float = 0.5):
    kind, D, C, H, T, K = CASES[name][:6]
    extra = CASES[name][6] if len(CASES[name]) > 6 else {}
    rng = np.random.default_rng(seed)
    perms = OF.random_perms(D, T, seed) if kind == "maf" else None
    st = dict(theta_mean=rng.normal(size=D).astype(np.float32),
              theta_std=rng.uniform(0.5, 2.0, size=D).astype(np.float32),
              x_mean=rng.normal(size=C).astype(np.float32),
              x_std=rng.uniform(0.5, 2.0, size=C).astype(np.float32))
    ospec = OF.FlowSpec(kind=kind, D=D, C=C, H=H, T=T, K=K, perms=perms, **extra,
                        **{k: v.astype(np.float64) for k, v in st.items()})
    spec = FlowSpec(kind=kind, D=D, C=C, H=H, T=T, K=K, perms=perms, **extra, **st)
    flat = OF.init_params(ospec, seed + 1)
    flat = (flat + spread * rng.normal(size=flat.shape) * np.abs(flat).mean()).astype(np.float32)
    theta = (rng.normal(size=(B, D)) * st["theta_std"] * 1.3 + st["theta_mean"]).astype(np.float32)
    x = (rng.normal(size=(B, C)) * st["x_std"] + st["x_mean"]).astype(np.float32)
    return ospec, spec, flat, theta, x


def oracle_log_prob(ospec, flat, theta, x, dtype=torch.float64):
    with torch.no_grad():
        return OF.log_prob(ospec, torch.as_tensor(flat).to(dtype), torch.as_tensor(theta).to(dtype),
                           torch.as_tensor(x).to(dtype)).double().numpy()


def oracle_inverse(ospec, flat, z, x, dtype=torch.float64):
    with torch.no_grad():
        th, ld = OF.inverse_transform(ospec, torch.as_tensor(flat).to(dtype), torch.as_tensor(z).to(dtype),
                                      torch.as_tensor(x).to(dtype))
    return th.double().numpy(), ld.double().numpy()

"""Generates tests/golden/*.npz: seeded inputs + expected outputs of the fp64 CPU oracle.

Provenance: these vectors come from THIS repository's restatement (oracle/), not from the reference's
stack (sbi / nflows are not installable here) -- they freeze the oracle so that neither it
nor the HIP path can drift unnoticed.  Re-run only when the specification itself changes:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cases import make_case, oracle_inverse, oracle_log_prob  # noqa: E402
from oracle import flows as OF  # noqa: E402
from oracle import philox  # noqa: E402

GOLDEN_CASES = ["maf_cfg1", "nsf_cfg3", "nsf_odd", "maf_small",
                # round 4: sbi's one-parameter NSF, the K = 10 coupling NSF, the autoregressive NSF of the lampe backend
                "nsf_d1", "nsf_k10", "nsfar_cfg1", "nsfar_small"]


def main():
    for name in GOLDEN_CASES:
        if os.path.exists(os.path.join(HERE, f"{name}.npz")) and "--all" not in sys.argv:
            continue   # frozen: existing vectors are only rewritten on request
        ospec, spec, flat, theta, x = make_case(name, seed=11, B=48)
        z = philox.normal(77, np.arange(48, dtype=np.uint64), 0, spec.D)
        lp = oracle_log_prob(ospec, flat, theta, x, torch.float64)
        th_inv, ld_inv = oracle_inverse(ospec, flat, z, x, torch.float64)
        p = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
        (-OF.log_prob(ospec, p, torch.as_tensor(theta).double(), torch.as_tensor(x).double())).mean().backward()
        np.savez_compressed(
            os.path.join(HERE, f"{name}.npz"), flat=flat, theta=theta, x=x, z=z, log_prob=lp, inv_theta=th_inv,
            inv_logdet=ld_inv, grad_mean_nll=p.grad.numpy().astype(np.float32),
            theta_mean=spec.theta_mean, theta_std=spec.theta_std, x_mean=spec.x_mean, x_std=spec.x_std,
            perms=spec.perms, meta=np.array([spec.D, spec.C, spec.H, spec.T, spec.K, spec.NB]), kind=spec.kind,
            tail_bound=np.float64(spec.tail_bound), ar_slope=np.float64(spec.ar_slope))
        print(name, "log_prob[:3]", lp[:3])


if __name__ == "__main__":
    main()

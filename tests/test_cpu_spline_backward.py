"""The hand-derived spline backward (numpy model of the HIP code) equals torch.autograd on the oracle."""
import numpy as np
import torch

from oracle import flows as OF
from spline_bwd_model import spline_fwd_bwd


def test_spline_backward_matches_autograd():
    rng = np.random.default_rng(0)
    for K in (4, 8, 10):
        spec = OF.FlowSpec(kind="nsf", D=2, C=1, H=50, T=1, K=K)
        for trial in range(40):
            q = rng.normal(size=3 * K - 1) * 4.0
            v = float(rng.uniform(-3.4, 3.4)) if trial % 5 else float(rng.choice([-3.0, 3.0, -2.999999, 0.0]))
            Go, Gl = rng.normal(), rng.normal()
            qt = torch.tensor(q[None, None, :], requires_grad=True)
            vt = torch.tensor([[v]], dtype=torch.float64, requires_grad=True)
            out, lad = OF.rq_spline(spec, vt, qt, inverse=False)
            (Go * out + Gl * lad).sum().backward()
            o, l, dv, dq = spline_fwd_bwd(q, v, Go, Gl, K, 50)
            assert abs(o - out.item()) < 1e-10 and abs(l - lad.item()) < 1e-9
            assert abs(dv - vt.grad.item()) < 1e-8 * max(1, abs(vt.grad.item())), (K, trial, dv, vt.grad.item())
            ref = qt.grad[0, 0].numpy()
            assert np.abs(dq - ref).max() < 1e-8 * max(1.0, np.abs(ref).max()), (K, trial)

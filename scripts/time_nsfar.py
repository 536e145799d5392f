"""Timings of the autoregressive NSF (kind "nsf_ar": the lampe / zuko backend's flow, csrc/sf_nsfar.hip) on the cfg1 parameter
space (D = 5, C = 10, H = 50, T = 5, 8 bins): log_prob, one training step's loss_grad, and the rejection sampler on a 2000 x 1000
catalogue inside a 3..97 % box.  Wall clock around synchronised calls."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
D, C = 5, 10
spec = FlowSpec(kind="nsf_ar", D=D, C=C, H=int(os.environ.get("SF_PROBE_H", "50")), T=5, K=8, tail_bound=5.0)
f = HipFlow(spec); flat = init_params(spec, g).cuda(); grad = torch.empty_like(flat)
f.set_params(flat)


def timed(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for B in (64, 16384, 131072):
    th = torch.randn(B, D, device="cuda"); x = torch.randn(B, C, device="cuda")
    print(f"nsf_ar B={B}: log_prob {timed(lambda: f.log_prob(th, x)):.3f} ms, loss_grad {timed(lambda: f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)):.3f} ms", flush=True)
f.set_params(flat)
M, S = 2000, 1000
x = torch.randn(M, C, device="cuda")
free = f.sample(x[:50], 400, seed=1).reshape(-1, D)
lo, hi = torch.quantile(free, 0.03, dim=0), torch.quantile(free, 0.97, dim=0)
out = torch.empty(M, S, D, device="cuda")
ms = timed(lambda: f.sample(x, S, lo, hi, seed=5, out=out), n=5)
print(f"nsf_ar sampler {M} x {S}: {ms:.2f} ms = {M * S / ms * 1e3:.3e} accepted draws/s, finite {bool(torch.isfinite(out).all())}")

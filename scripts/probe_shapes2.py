"""log_prob and loss_grad across MAF / NSF shapes (diagnostics)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
B, C = 262144, 10
def t(fn, n=4):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
for kind in ("maf", "nsf"):
    for H in (50, 64, 100):
        for D in (5, 8):
            spec = FlowSpec(kind=kind, D=D, C=C, H=H, T=5, K=8, perms=random_perms(D, 5, g) if kind == "maf" else None)
            f = HipFlow(spec); flat = init_params(spec, g).cuda(); f.set_params(flat)
            th = torch.randn(B, D, device="cuda"); x = torch.randn(B, C, device="cuda"); grad = torch.empty_like(flat)
            d = f.describe()
            lp = t(lambda: f.log_prob(th, x)); lg = t(lambda: f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad))
            xs = torch.randn(2000, C, device="cuda"); out = torch.empty(2000, 131, D, device="cuda")
            sm = t(lambda: f.sample(xs, 131, seed=1, out=out))
            print(f"{kind} H={H} D={D}: HT={d['HT']} n_parts={d['n_parts']}  log_prob {lp:.2f} ms  loss_grad {lg:.2f} ms  sample(262k) {sm:.2f} ms")

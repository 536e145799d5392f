"""Training-step timing vs batch (diagnostics)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
for kind, D, C, K, flp in [("maf", 5, 10, 10, 40030.0), ("nsf", 8, 20, 8, 178640.0)]:
    spec = FlowSpec(kind=kind, D=D, C=C, H=50, T=5, K=K, perms=random_perms(D, 5, g) if kind == "maf" else None)
    f = HipFlow(spec); flat = init_params(spec, g).cuda(); grad = torch.empty_like(flat)
    for B in (64, 2048, 16384, 65536, 262144):
        th = torch.randn(B, D, device="cuda"); x = torch.randn(B, C, device="cuda")
        for _ in range(3): f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20 if B <= 65536 else 5
        for _ in range(n): f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{kind} loss_grad B={B:7d}: {dt*1e3:8.3f} ms  {B/dt/1e6:8.1f} Mpairs/s  {3*flp*B/dt/1e12:6.2f} TF (3x logprob flops)")

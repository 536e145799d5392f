"""Training step time for wide flows (HT = 3, 4) (diagnostics)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
for kind, D, C, H, T, K in [("nsf", 8, 20, 69, 15, 10), ("nsf", 8, 20, 100, 5, 8), ("maf", 8, 20, 69, 5, 10), ("maf", 8, 20, 128, 5, 10)]:
    spec = FlowSpec(kind=kind, D=D, C=C, H=H, T=T, K=K, perms=random_perms(D, T, g) if kind == "maf" else None)
    f = HipFlow(spec); flat = init_params(spec, g).cuda(); grad = torch.empty_like(flat)
    for B in (64, 16384, 131072):
        th = torch.randn(B, D, device="cuda"); x = torch.randn(B, C, device="cuda")
        for _ in range(2): f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
        print(f"{kind} H={H} T={T} HT={f.describe()['HT']} B={B}: {dt*1e3:.2f} ms  {B/dt/1e6:.2f} Mpairs/s")

"""One training configuration per run (diagnostics under rocprofv3). SF_PROBE_KIND=maf|nsf, SF_PROBE_B=batch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
kind = os.environ.get("SF_PROBE_KIND", "maf"); B = int(os.environ.get("SF_PROBE_B", "65536"))
D, C, K = (5, 10, 10) if kind == "maf" else (8, 20, 8)
spec = FlowSpec(kind=kind, D=D, C=C, H=50, T=5, K=K, perms=random_perms(D, 5, g) if kind == "maf" else None)
f = HipFlow(spec); flat = init_params(spec, g).cuda(); grad = torch.empty_like(flat)
th = torch.randn(B, D, device="cuda"); x = torch.randn(B, C, device="cuda")
for _ in range(2): f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("SF_PROBE_N", "5"))
for _ in range(n): f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"{kind} loss_grad B={B}: {dt*1e3:.3f} ms {B/dt/1e6:.1f} Mpairs/s")

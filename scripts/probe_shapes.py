"""Dense MAF sampling round for several (D, H) shapes (diagnostics)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
M, S, C = 2000, 1000, 10
for H in (50, 64):
    for D in (3, 4, 5, 6, 7, 8, 10, 12):
        spec = FlowSpec(kind="maf", D=D, C=C, H=H, T=5, perms=random_perms(D, 5, g))
        f = HipFlow(spec); f.set_params(init_params(spec, g))
        x = torch.randn(M, C, device="cuda"); out = torch.empty(M, S, D, device="cuda")
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f.sample(x, S, seed=1, out=out); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        d = f.describe()
        print(f"H={H} D={D}: m16={d['m16_ok']} nT16={d['nT16']}  {min(ts)*1e3:.2f} ms")

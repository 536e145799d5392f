"""MAF cfg2-shaped dense sampling round only (diagnostics under rocprofv3)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow

g = torch.Generator().manual_seed(0)
D, C, M, S = 5, 10, 2000, 1000
spec = FlowSpec(kind="maf", D=D, C=C, H=50, T=5, perms=random_perms(D, 5, g))
f = HipFlow(spec); f.set_params(init_params(spec, g))
x = torch.randn(M, C, device="cuda"); out = torch.empty(M, S, D, device="cuda")
n = int(os.environ.get("SF_PROBE_N", "5"))
ts = []
for _ in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f.sample(x, S, seed=1, out=out); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"maf sample M={M} S={S}: min {min(ts)*1e3:.2f} ms")

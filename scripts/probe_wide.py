import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/synference_amd") else os.getcwd())
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
M, S, C = 500, 1000, 10
for H in (69, 100, 128):
    for D in (3, 5, 8, 12):
        spec = FlowSpec(kind="maf", D=D, C=C, H=H, T=5, perms=random_perms(D, 5, g))
        f = HipFlow(spec); f.set_params(init_params(spec, g))
        x = torch.randn(M, C, device="cuda"); out = torch.empty(M, S, D, device="cuda")
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f.sample(x, S, seed=1, out=out); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        d = f.describe()
        print(f"H={H} D={D}: HT={d['HT']} inc_ok={d['inc_ok']} n_parts={d['n_parts']}  {min(ts)*1e3:.2f} ms per 5e5 draws = {min(ts)/(M*S)*1e9:.1f} ns/draw")

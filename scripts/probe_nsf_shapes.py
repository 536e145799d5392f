"""NSF sampler / log_prob time per draw across shapes (diagnostics: looks for performance cliffs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
M, S, C = 500, 1000, 20
for H in (32, 50, 64, 69, 100, 128):
    for D, K in ((3, 8), (8, 8), (8, 16), (16, 10)):
        spec = FlowSpec(kind="nsf", D=D, C=C, H=H, T=5, K=K)
        f = HipFlow(spec); f.set_params(init_params(spec, g))
        x = torch.randn(M, C, device="cuda"); out = torch.empty(M, S, D, device="cuda")
        th = torch.randn(M * S, D, device="cuda"); xx = x.repeat_interleave(S, 0)
        def t(fn):
            fn(); ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return min(ts) / (M * S) * 1e9
        d = f.describe()
        print(f"H={H} D={D} K={K}: HT={d['HT']} PT={d['PT']} n_parts={d['n_parts']}  sample {t(lambda: f.sample(x, S, seed=1, out=out)):.1f} ns/draw  log_prob {t(lambda: f.log_prob(th, xx)):.1f} ns/row")

"""Quick GPU timing probe of the hot kernels at BASELINE shapes (diagnostics, not the bench)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    return min(t), float(np.median(t))

g = torch.Generator().manual_seed(0)
BF = os.environ.get("SF_PROBE_BF16") == "1"
for kind, D, C, K, M, S in [("maf", 5, 10, 10, 2000, 1000), ("nsf", 8, 20, 8, 2000, 1000)]:
    spec = FlowSpec(kind=kind, D=D, C=C, H=50, T=5, K=K, perms=random_perms(D, 5, g) if kind == "maf" else None,
                    hidden_bf16=BF)
    f = HipFlow(spec); flat = init_params(spec, g); f.set_params(flat)
    x = torch.randn(M, C, device="cuda"); th = torch.randn(M * 100, D, device="cuda"); xx = x.repeat_interleave(100, 0)
    out = torch.empty(M, S, D, device="cuda")
    tmin, tmed = timeit(lambda: f.sample(x, S, seed=1, out=out))
    print(f"{kind} sample  M={M} S={S}: min {tmin*1e3:.2f} ms  -> {M*S/tmin/1e6:.1f} Msamples/s")
    tmin, tmed = timeit(lambda: f.log_prob(th, xx))
    print(f"{kind} logprob B={th.shape[0]}: min {tmin*1e3:.2f} ms  -> {th.shape[0]/tmin/1e6:.1f} Mrows/s")

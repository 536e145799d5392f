"""One dense sampling round per launch (diagnostics under rocprofv3). SF_PROBE_KIND=maf|nsf|nsfprod, SF_PROBE_OP=sample|logprob."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow

g = torch.Generator().manual_seed(0)
KIND = os.environ.get("SF_PROBE_KIND", "maf")
OP = os.environ.get("SF_PROBE_OP", "sample")
if KIND == "maf":
    D, C, M, S = 5, 10, 2000, 1000
    spec = FlowSpec(kind="maf", D=D, C=C, H=50, T=5, perms=random_perms(D, 5, g))
elif KIND == "nsf":
    D, C, M, S = 8, 20, 2000, 1000
    spec = FlowSpec(kind="nsf", D=D, C=C, H=50, T=5, K=8)
else:
    D, C, M, S = 8, 20, 1000, 1000
    spec = FlowSpec(kind="nsf", D=D, C=C, H=69, T=15, K=10)
f = HipFlow(spec); f.set_params(init_params(spec, g))
x = torch.randn(M, C, device="cuda"); out = torch.empty(M, S, D, device="cuda")
n = int(os.environ.get("SF_PROBE_N", "5"))
ts = []
th = torch.randn(M * S, D, device="cuda"); xx = x.repeat_interleave(S, 0)
for _ in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if OP == "sample":
        f.sample(x, S, seed=1, out=out)
    else:
        f.log_prob(th, xx)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"{KIND} {OP} M={M} S={S}: min {min(ts)*1e3:.2f} ms")

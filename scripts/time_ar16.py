"""Sampler timings of the lampe-backend flows (sf_nsfar16.hip vs SF_AR_SAMP16=0: k_ar_sample): a random-weight flow of the bench
shape, 2000 rows x 1000 draws, (a) no prior box -- every candidate accepted, exactly M x S evaluations, no tail -- and (b) a box
from the 3..97 % quantiles.  Prints the persistent launch's HIP-event time, evaluations and rounds."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
KIND = os.environ.get("SF_PROBE_KIND", "nsf_ar")
D, C, H, T, K = (int(os.environ.get("SF_PROBE_" + k, v)) for k, v in (("D", 5), ("C", 10), ("H", 50), ("T", 5), ("K", 8)))
M, S = int(os.environ.get("SF_PROBE_M", 2000)), int(os.environ.get("SF_PROBE_S", 1000))
spec = FlowSpec(kind=KIND, D=D, C=C, H=H, T=T, K=K, tail_bound=5.0)
f = HipFlow(spec); flat = init_params(spec, g).cuda()
flat = flat + 0.3 * flat.abs().mean() * torch.randn(flat.shape, generator=g).cuda()
f.set_params(flat); f.set_profiling(True)
X = torch.randn(M, C, device="cuda")
out = torch.empty(M, S, D, device="cuda")
free = f.sample(X[:64], 512, seed=1).reshape(-1, D)
lo = torch.quantile(free, 0.03, dim=0).cpu().numpy(); hi = torch.quantile(free, 0.97, dim=0).cpu().numpy()
big = 1e30 * np.ones(D, np.float32)
for name, box, kw in (("no box", (None, None), {}), ("box that accepts everything", (-big, big), {}), ("box 3..97 %", (lo, hi), {}),
                      ("box 3..97 %, max_attempts 1000", (lo, hi), dict(max_attempts=1000))):
    for _ in range(2): f.sample(X, S, *box, seed=5, out=out, **kw)
    ks, ws, ev, rd = [], [], 0, 0
    for k in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f.sample(X, S, *box, seed=10 + k, out=out, **kw)
        torch.cuda.synchronize(); ws.append(time.perf_counter() - t0)
        st = f.last_sample_stats; ks.append(st["dense_ms"]); ev = st["evaluations"]; rd = st["rounds"]
    print(f"{KIND} D{D} C{C} H{H} T{T} K{K} {name}: launch {np.median(ks):.3f} ms, call {1e3*np.median(ws):.3f} ms, evaluations {ev:.0f} "
          f"({ev/(M*S):.3f} per draw), rounds {rd}, {M*S/np.median(ws)/1e6:.1f} M draws/s", flush=True)
# acceptance counting: one attempt per item, the box test, no retries and no draws written
for _ in range(2): f.acceptance(X, S, lo, hi, seed=3)
ws = []
for k in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    acc = f.acceptance(X, S, lo, hi, seed=20 + k)
    torch.cuda.synchronize(); ws.append(time.perf_counter() - t0)
a_ = acc.float().cpu().numpy()
print(f"acceptance counting, box 3..97 %: call {1e3*np.median(ws):.3f} ms per {M*S} evaluations; acceptance per row: mean {a_.mean():.3f}, "
      f"min {a_.min():.4f}, 1 % quantile {np.quantile(a_, 0.01):.4f}, rows below 0.1: {(a_ < 0.1).sum()}", flush=True)

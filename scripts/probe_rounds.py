"""Per-round timing of the rejection sampler (diagnostics)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.estimator import build_flow
from synference_amd.priors import prior_from_parameters
from synference_amd.runner import HipAdam
from synference_amd.synthetic import make_catalogue

kind = sys.argv[1] if len(sys.argv) > 1 else "maf"
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
D, C, K, NL, M = (5, 10, 10, 10000, 2000) if kind == "maf" else (8, 20, 8, 100000, 20000)
dev = torch.device("cuda:0")
x_lib, th_lib, names = make_catalogue(NL, C, D, seed=1234)
x_test, _, _ = make_catalogue(M, C, D, seed=4321)
tr = np.random.RandomState(0).permutation(NL)[: int(0.8 * NL)]
prior = prior_from_parameters(th_lib[tr], names)
est = build_flow(kind, th_lib[tr], x_lib[tr], hidden_features=50, num_transforms=5, num_bins=K, device=dev,
                 generator=torch.Generator().manual_seed(42)).to(dev)
flow, flat = est.flow, est.flat.data
Xtr, Ttr = torch.as_tensor(x_lib[tr]).to(dev), torch.as_tensor(th_lib[tr], dtype=torch.float32).to(dev)
grad = torch.empty_like(flat); opt = HipAdam(flat, lr=2e-3); g2 = torch.Generator().manual_seed(7)
for it in range(4000):
    bi = torch.randint(0, len(tr), (2048,), generator=g2).to(dev)
    opt.desc.lr = 2e-3 * 0.5 * (1 + np.cos(np.pi * it / 4000))
    flow.loss_grad(flat, Ttr[bi], Xtr[bi], 1 / 2048, grad_out=grad); opt.step(grad, 5.0)
flow.set_params(flat)
lo, hi = prior.low.to(dev), prior.high.to(dev)
X = torch.as_tensor(x_test).to(dev); S = 1000
out = torch.empty((M, S, D), device=dev); rej = [torch.empty(M * S, dtype=torch.int32, device=dev) for _ in range(2)]
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
for rep in range(2):
    pending, cur, attempt, r = M * S, None, 0, 0
    rows = []
    while pending > 0 and attempt < 64:
        A = 1
        if attempt > 0:
            while A < 16 and 2 * A * pending <= budget and attempt + 2 * A <= 64: A *= 2
        cnt.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        flow.sample_round(X, S, cur, 0, pending, attempt, 5, lo, hi, out, rej[r & 1], cnt, attempts_per_slot=A)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nxt = int(cnt.item()); rows.append((attempt, A, pending, pending * A, round(dt * 1e3, 3), nxt))
        pending, cur, attempt, r = nxt, rej[r & 1], attempt + A, r + 1
print("attempt A pending items ms next_pending")
for row in rows: print(*row)
print("total ms", sum(r[4] for r in rows))

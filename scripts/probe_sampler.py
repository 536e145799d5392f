"""GPU probe: cost of the sampler's attempt ceilings on the bench workload (under-trained mock posterior)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.estimator import build_flow
from synference_amd.priors import prior_from_parameters
from synference_amd.runner import HipAdam
from synference_amd.synthetic import make_catalogue

kind = sys.argv[1] if len(sys.argv) > 1 else "maf"
D, C, K, nlib, M = (5, 10, 10, 10000, 2000) if kind == "maf" else (8, 20, 8, 100000, 20000)
dev = torch.device("cuda:0")
x_lib, th_lib, names = make_catalogue(nlib, C, D, seed=1234)
x_test, th_test, _ = make_catalogue(M, C, D, seed=4321)
idx = np.random.RandomState(0).permutation(len(x_lib)); tr = idx[: int(0.8 * len(idx))]
prior = prior_from_parameters(th_lib[tr], names)
est = build_flow(kind, th_lib[tr], x_lib[tr], hidden_features=50, num_transforms=5, num_bins=K, device=dev,
                 generator=torch.Generator().manual_seed(42)).to(dev)
flow, flat = est.flow, est.flat.data
Xtr = torch.as_tensor(x_lib[tr]).to(dev); Ttr = torch.as_tensor(th_lib[tr], dtype=torch.float32).to(dev)
grad = torch.empty_like(flat); opt = HipAdam(flat, lr=1e-3); g2 = torch.Generator().manual_seed(7)
for it in range(4000):
    bi = torch.randint(0, len(tr), (2048,), generator=g2).to(dev)
    opt.desc.lr = 2e-3 * 0.5 * (1.0 + np.cos(np.pi * it / 4000))
    flow.loss_grad(flat, Ttr[bi], Xtr[bi], 1.0 / 2048, grad_out=grad); opt.step(grad, 5.0)
flow.set_params(flat)
lo, hi = prior.low.to(dev), prior.high.to(dev)
X = torch.as_tensor(x_test).to(dev); S = 1000
out = torch.empty((M, S, D), dtype=torch.float32, device=dev)
for cap in (64, 1024, 16384, 262144, None):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o, nd = flow.sample(X, S, lo, hi, seed=1000 + rep, max_attempts=cap, out=out, return_counts=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = flow.last_sample_stats
    acc = (S / nd.float().clamp_min(1))
    print(f"{kind} cap={cap}: {dt*1e3:.2f} ms/step  kernel1={st['dense_ms']:.3f} ms launches={st['rounds']} unfilled={flow.last_unfilled} "
          f"evals={st['evaluations']:.3e} min_acc={float(acc.min()):.2e} n(acc<1%)={int((acc<0.01).sum())} n(acc<1e-3)={int((acc<1e-3).sum())}", flush=True)

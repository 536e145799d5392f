"""GPU probe: where does the bench spend its time / hang?  Prints a line per phase."""
import faulthandler, sys, time, os
faulthandler.dump_traceback_later(70, repeat=True)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t00 = time.perf_counter()
def note(m): print(f"[{time.perf_counter()-t00:7.2f}s] {m}", flush=True)
from synference_amd.estimator import build_flow
from synference_amd.priors import prior_from_parameters
from synference_amd.runner import HipAdam
from synference_amd.synthetic import make_catalogue
D, C, K, nlib, M = 5, 10, 10, 10000, 2000
dev = torch.device("cuda:0")
x_lib, th_lib, names = make_catalogue(nlib, C, D, seed=1234)
x_test, th_test, _ = make_catalogue(M, C, D, seed=4321)
idx = np.random.RandomState(0).permutation(len(x_lib)); tr = idx[: int(0.8 * len(idx))]
prior = prior_from_parameters(th_lib[tr], names)
est = build_flow("maf", th_lib[tr], x_lib[tr], hidden_features=50, num_transforms=5, num_bins=K, device=dev,
                 generator=torch.Generator().manual_seed(42)).to(dev)
flow, flat = est.flow, est.flat.data
note("flow built")
Xtr = torch.as_tensor(x_lib[tr]).to(dev); Ttr = torch.as_tensor(th_lib[tr], dtype=torch.float32).to(dev)
grad = torch.empty_like(flat); opt = HipAdam(flat, lr=1e-3); g2 = torch.Generator().manual_seed(7)
nfit = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for it in range(nfit):
    bi = torch.randint(0, len(tr), (2048,), generator=g2).to(dev)
    opt.desc.lr = 2e-3 * 0.5 * (1.0 + np.cos(np.pi * it / nfit))
    flow.loss_grad(flat, Ttr[bi], Xtr[bi], 1.0 / 2048, grad_out=grad); opt.step(grad, 5.0)
    if it in (0, 9, 99, 999): torch.cuda.synchronize(); note(f"fit step {it+1}")
torch.cuda.synchronize(); note("fit done")
flow.set_params(flat)
lo, hi = prior.low.to(dev), prior.high.to(dev)
X = torch.as_tensor(x_test).to(dev); S = 1000
for m in (2000,):
    for cap in (1, 64, 1024, None):
        out = torch.empty((m, S, D), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o, nd = flow.sample(X[:m], S, lo, hi, seed=1000, max_attempts=cap, out=out, return_counts=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = flow.last_sample_stats
        acc = (S / nd.float().clamp_min(1)).sort().values
        note(f"lowest acceptances: {[round(float(v), 5) for v in acc[:6]]}")
        note(f"sample M={m} cap={cap}: {dt*1e3:.2f} ms kernel={st['dense_ms']:.3f} ms launches={st['rounds']} unfilled={flow.last_unfilled} evals={st['evaluations']:.3e}")

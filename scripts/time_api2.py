"""Host-side breakdown of SBI_Fitter.sample_posterior on the bench workload (MAF cfg2 shape, untrained flow widened a little)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd import hostio
from synference_amd.estimator import build_flow
from synference_amd.fitter import SBI_Fitter
from synference_amd.posterior import EnsemblePosterior, FlowPosterior
from synference_amd.priors import prior_from_parameters
from synference_amd.synthetic import make_catalogue

dev = torch.device("cuda:0")
x, th, names = make_catalogue(10000, 10, 5, seed=1234)
est = build_flow("maf", th[:8000], x[:8000], hidden_features=50, num_transforms=5, device=dev, generator=torch.Generator().manual_seed(1)).to(dev)
prior = prior_from_parameters(th[:8000], names)
# a short seeded fit (as bench.py's warm-up fit): an untrained flow is accepted by the prior box once in thousands of draws
from synference_amd.runner import HipAdam
flat = est.flat.data; grad = torch.empty_like(flat); opt = HipAdam(flat, lr=1e-3)
Ttr = torch.as_tensor(th[:8000], dtype=torch.float32).to(dev); Xtr = torch.as_tensor(x[:8000]).to(dev)
g2 = torch.Generator().manual_seed(7)
for it in range(3000):
    bi = torch.randint(0, 8000, (2048,), generator=g2).to(dev)
    opt.desc.lr = 2e-3 * 0.5 * (1.0 + np.cos(np.pi * it / 3000))
    est.flow.loss_grad(flat, Ttr[bi], Xtr[bi], 1.0 / 2048, grad_out=grad)
    opt.step(grad, 5.0)
est.flow.set_params(flat)
post = FlowPosterior(est, prior)
est._sync_params()
fit = SBI_Fitter("t", names, [f"F{i}" for i in range(10)], feature_array=x, parameter_array=th)
fit.posteriors = EnsemblePosterior([post], weights=[1.0]); fit._prior = prior
X = x[8000:10000]; S = 1000
Xd = torch.as_tensor(X).to(dev)
lo, hi = prior.low.to(dev), prior.high.to(dev)
out = torch.empty((2000, S, 5), device=dev)


def t(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n


print(f"flow.sample (engine call, device out)     {t(lambda: est.flow.sample(Xd, S, lo, hi, seed=3, out=out)):.3f} ms")
print(f"FlowPosterior.sample_catalogue            {t(lambda: post.sample_catalogue(Xd, S, 3)):.3f} ms")
print(f"EnsemblePosterior.sample_catalogue (np X) {t(lambda: fit.posteriors.sample_catalogue(torch.as_tensor(X), S, 3)):.3f} ms")
print(f"to_host_f64 alone                         {t(lambda: hostio.to_host_f64(out)):.3f} ms")
print(f"SBI_Fitter.sample_posterior               {t(lambda: fit.sample_posterior(X, num_samples=S, seed=3)):.3f} ms")
import os
ref = fit.sample_posterior(X, num_samples=S, seed=3).copy()
for nc in (1, 2, 3, 4, 6):
    os.environ["SF_API_CHUNKS"] = str(nc)
    same = np.array_equal(fit.sample_posterior(X, num_samples=S, seed=3), ref, equal_nan=True)
    print(f"SBI_Fitter.sample_posterior, {nc} chunk(s)    {t(lambda: fit.sample_posterior(X, num_samples=S, seed=3)):.3f} ms   identical draws: {same}")
os.environ.pop("SF_API_CHUNKS")
if os.environ.get("SF_API_PROFILE"):
    import cProfile, pstats, io
    pr = cProfile.Profile()
    for _ in range(3): fit.sample_posterior(X, num_samples=S, seed=3)
    pr.enable()
    for _ in range(20): fit.sample_posterior(X, num_samples=S, seed=3)
    pr.disable()
    sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(28); print(sio.getvalue()[:6000])

"""Per-galaxy acceptance of the default bench catalogue (the shape of the sampler's retry work) + an idealised
simulation of the rejection schedule: how long the tail has to be when every lane of the chip is used."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from synference_amd.estimator import build_flow
from synference_amd.priors import prior_from_parameters
from synference_amd.runner import HipAdam
from synference_amd.synthetic import make_catalogue
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "maf_cfg2"]
D, C = wl["D"], wl["C"]
dev = torch.device("cuda", 0)
x_lib, th_lib, names = make_catalogue(wl["n_lib"], C, D, seed=1234)
x_all, th_all, _ = make_catalogue(wl["galaxies"], C, D, seed=4321)
rs = np.random.RandomState(0); idx = rs.permutation(len(x_lib)); tr = idx[: int(0.8 * len(idx))]
prior = prior_from_parameters(th_lib[tr], names)
gen = torch.Generator().manual_seed(42)
est = build_flow(wl["kind"], th_lib[tr], x_lib[tr], hidden_features=wl.get("H", 50), num_transforms=wl.get("T", 5),
                 num_bins=wl["K"], device=dev, generator=gen).to(dev)
flow, flat = est.flow, est.flat.data
Xtr = torch.as_tensor(x_lib[tr]).to(dev); Ttr = torch.as_tensor(th_lib[tr], dtype=torch.float32).to(dev)
grad = torch.empty_like(flat); opt = HipAdam(flat, lr=1e-3); g2 = torch.Generator().manual_seed(7)
for it in range(int(os.environ.get("FIT_STEPS", "4000"))):
    bi = torch.randint(0, len(tr), (2048,), generator=g2).to(dev)
    opt.desc.lr = float(os.environ.get("FIT_LR", "2e-3")) * 0.5 * (1.0 + np.cos(np.pi * it / 4000))
    flow.loss_grad(flat, Ttr[bi], Xtr[bi], 1.0 / 2048, grad_out=grad); opt.step(grad, 5.0)
flow.set_params(flat)
X = torch.as_tensor(x_all).to(dev)
acc = flow.acceptance(X, 20000, prior.low.to(dev), prior.high.to(dev), seed=5).cpu().numpy()   # (fractions)
print("galaxies", len(acc), "mean acceptance", acc.mean(), "expected attempts per slot", (1 / np.maximum(acc, 1e-5)).mean())
qs = [0, 0.001, 0.01, 0.05, 0.1, 0.25, 0.5]
print("quantiles of p:", {q: float(np.quantile(acc, q)) for q in qs})
w = 1 / np.maximum(acc, 1e-5) - 1
order = np.argsort(acc)
print("share of all RETRY attempts owed to the hardest 1 % / 5 % / 10 % of galaxies:",
      [float(w[order[: int(f * len(acc))]].sum() / w.sum()) for f in (0.01, 0.05, 0.1)])
np.save("gpurun_out/acceptance_%s.npy" % (sys.argv[1] if len(sys.argv) > 1 else "maf_cfg2"), acc)

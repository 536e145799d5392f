"""Developer probe: how far apart are two ALL-fp32 evaluations of the same draws?  The bench flow (MAF cfg1, fitted) samples 256
galaxies x 1000 draws with the same seed through (a) the 16-row fp32 sampler, (b) the split-bf16 x3 fast mode, and -- in a second
process started with SF_MAF16=0 -- (c) the 32-row fp32 kernels; |d log_prob| of the draws under the fp32 density kernel, the
statistic of bench.py's split_bf16_cost.  (c) vs (a) is the floor any alternative arithmetic is measured against.
    python scripts/mode_noise.py write OUT.npz          (run twice: plain, and with SF_MAF16=0)
    python scripts/mode_noise.py compare A.npz B.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build():
    from synference_amd.estimator import build_flow
    from synference_amd.priors import prior_from_parameters
    from synference_amd.runner import HipAdam
    from synference_amd.synthetic import make_catalogue
    dev = torch.device("cuda:0")
    x_lib, th_lib, names = make_catalogue(10_000, 10, 5, seed=1234)
    x_all, _, _ = make_catalogue(2000, 10, 5, seed=4321)
    idx = np.random.RandomState(0).permutation(len(x_lib))
    tr = idx[:8000]
    prior = prior_from_parameters(th_lib[tr], names)
    est = build_flow("maf", th_lib[tr], x_lib[tr], hidden_features=50, num_transforms=5, num_bins=10, device=dev,
                     generator=torch.Generator().manual_seed(42)).to(dev)
    flow, flat = est.flow, est.flat.data
    Xtr, Ttr = torch.as_tensor(x_lib[tr]).to(dev), torch.as_tensor(th_lib[tr], dtype=torch.float32).to(dev)
    grad = torch.empty_like(flat)
    opt = HipAdam(flat, lr=1e-3)
    g2 = torch.Generator().manual_seed(7)
    for it in range(4000):
        bi = torch.randint(0, len(tr), (2048,), generator=g2).to(dev)
        opt.desc.lr = 2e-3 * 0.5 * (1.0 + np.cos(np.pi * it / 4000))
        flow.loss_grad(flat, Ttr[bi], Xtr[bi], 1.0 / 2048, grad_out=grad)
        opt.step(grad, 5.0)
    flow.set_params(flat)
    return est, flow, prior, torch.as_tensor(x_all[:256]).to(dev), dev


def main():
    if sys.argv[1] == "write":
        from synference_amd import _lib
        est, flow, prior, X, dev = build()
        lo, hi = prior.low.to(dev), prior.high.to(dev)
        out = {}
        for mode in (1, 0):
            _lib.load().sf_set_sampler_fp32(mode)
            o = torch.empty((256, 1000, 5), dtype=torch.float32, device=dev)
            flow.sample(X, 1000, lo, hi, seed=4242, out=o)
            xr = X.repeat_interleave(1000, 0)
            out[f"theta{mode}"] = o.cpu().numpy()
            out[f"lp{mode}"] = flow.log_prob(o.reshape(-1, 5), xr).reshape(256, 1000).cpu().numpy()
        out["sigma"] = np.asarray(est.spec.theta_std)
        np.savez(sys.argv[2], **out)
        return
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])

    def cmp(t1, l1, t2, l2, what):
        dth = (np.abs(t1 - t2) / a["sigma"]).max(-1)
        same = dth < 1e-3
        d = np.abs(l1 - l2)[same]
        print(f"{what}: compared {same.sum()}, differ {np.sum(~same)}, max |dlogp| {d.max():.3e}, p99.9 {np.quantile(d, 0.999):.3e}, "
              f"median {np.median(d):.3e}, max dtheta/sigma {dth[same].max():.3e}")
    cmp(a["theta1"], a["lp1"], a["theta0"], a["lp0"], "A fp32 vs A split x3")
    cmp(a["theta1"], a["lp1"], b["theta1"], b["lp1"], "A fp32 vs B fp32 (two all-fp32 kernels)")
    cmp(b["theta1"], b["lp1"], a["theta0"], a["lp0"], "B fp32 vs A split x3")


if __name__ == "__main__":
    main()

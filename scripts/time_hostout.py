"""Developer probe: the sampler writing its draws straight into PINNED HOST memory (the kernel's stores cross PCIe while it runs)
against device output + a D2H copy afterwards.  Prints kernel / wall times per catalogue call of the bench workload."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from mode_noise import build  # noqa: E402


def main():
    est, flow, prior, _, dev = build()
    from synference_amd.synthetic import make_catalogue
    x_all, _, _ = make_catalogue(2000, 10, 5, seed=4321)
    X = torch.as_tensor(x_all).to(dev)
    lo, hi = prior.low.to(dev), prior.high.to(dev)
    out_d = torch.empty((2000, 1000, 5), dtype=torch.float32, device=dev)
    out_h = torch.empty((2000, 1000, 5), dtype=torch.float32, pin_memory=True)
    # (engine.sample takes any tensor whose data_ptr the device can address: pinned host memory is mapped into the GPU's space)
    out_d64 = torch.empty((2000, 1000, 5), dtype=torch.float64, device=dev)
    out_h64 = torch.empty((2000, 1000, 5), dtype=torch.float64, pin_memory=True)
    for name, out in (("device", out_d), ("pinned host", out_h), ("device f64", out_d64), ("pinned host f64", out_h64), ("device", out_d),
                      ("pinned host", out_h), ("pinned host f64", out_h64)):
        ks, ws = [], []
        for k in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            flow.sample(X, 1000, lo, hi, seed=100 + k, out=out)
            torch.cuda.synchronize()
            ws.append(time.perf_counter() - t0)
            ks.append(flow.last_sample_stats["dense_ms"])
        print(f"{name:12s}: wall {1e3 * np.median(ws):.3f} ms, persistent launch {np.median(ks):.3f} ms, unfilled {flow.last_unfilled}")
    # the API call itself, both ways
    from synference_amd.fitter import SBI_Fitter
    from synference_amd.posterior import EnsemblePosterior, FlowPosterior
    fitter = SBI_Fitter("t", ["a", "b", "c", "d", "e"], [f"F{i}" for i in range(10)], feature_array=x_all, parameter_array=np.zeros((2000, 5)))
    fitter.posteriors = EnsemblePosterior([FlowPosterior(est, prior)], weights=[1.0])
    fitter._prior = prior
    for direct in ("1", "0", "1", "0"):
        os.environ["SF_API_DIRECT"] = direct
        ts = []
        for k in range(8):
            t0 = time.perf_counter()
            arr = fitter.sample_posterior(x_all, num_samples=1000, seed=300 + k, shard=False)
            ts.append(time.perf_counter() - t0)
        print(f"sample_posterior SF_API_DIRECT={direct}: median {1e3 * np.median(ts):.3f} ms, min {1e3 * np.min(ts):.3f}")
    flow.sample(X, 1000, lo, hi, seed=5, out=out_d)
    flow.sample(X, 1000, lo, hi, seed=5, out=out_h)
    torch.cuda.synchronize()
    print("identical:", bool(torch.equal(out_d.cpu(), out_h)))


if __name__ == "__main__":
    main()

"""Developer probe: cProfile of SBI_Fitter.sample_posterior on the bench workload (where the host-side 0.4 ms go)."""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from mode_noise import build  # noqa: E402


def main():
    est, flow, prior, _, dev = build()
    from synference_amd.fitter import SBI_Fitter
    from synference_amd.posterior import EnsemblePosterior, FlowPosterior
    from synference_amd.synthetic import make_catalogue
    x_all, _, _ = make_catalogue(2000, 10, 5, seed=4321)
    fitter = SBI_Fitter("t", ["a", "b", "c", "d", "e"], [f"F{i}" for i in range(10)], feature_array=x_all, parameter_array=np.zeros((2000, 5)))
    fitter.posteriors = EnsemblePosterior([FlowPosterior(est, prior)], weights=[1.0])
    fitter._prior = prior
    for k in range(5):
        arr = fitter.sample_posterior(x_all, num_samples=1000, seed=k, shard=False)
    pr = cProfile.Profile()
    pr.enable()
    for k in range(100):
        arr = fitter.sample_posterior(x_all, num_samples=1000, seed=300 + k, shard=False)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()

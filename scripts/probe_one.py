"""GPU probe: one sampler configuration, errors printed not raised."""
import faulthandler, sys, time, os
faulthandler.dump_traceback_later(50, exit=True)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.estimator import build_flow
from synference_amd.priors import prior_from_parameters
from synference_amd.synthetic import make_catalogue
D, C, K, nlib = 5, 10, 10, 10000
dev = torch.device("cuda:0")
x_lib, th_lib, names = make_catalogue(nlib, C, D, seed=1234)
x_test, th_test, _ = make_catalogue(2000, C, D, seed=4321)
prior = prior_from_parameters(th_lib, names)
est = build_flow("maf", th_lib, x_lib, hidden_features=50, num_transforms=5, num_bins=K, device=dev,
                 generator=torch.Generator().manual_seed(42)).to(dev)
flow = est.flow
flow.set_params(est.flat.data)
lo, hi = prior.low.to(dev), prior.high.to(dev)
X = torch.as_tensor(x_test).to(dev); S = 1000
tag = os.environ.get("TAG", "")
for m, cap in [(4, 8), (16, 8), (64, 2), (64, 8), (500, 8), (2000, 64)]:
    out = torch.empty((m, S, D), dtype=torch.float32, device=dev)
    try:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        flow.sample(X[:m], S, lo, hi, seed=1000, max_attempts=cap, out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = flow.last_sample_stats
        print(f"{tag} M={m} cap={cap}: {dt*1e3:.2f} ms kernel={st['dense_ms']:.3f} launches={st['rounds']} unfilled={flow.last_unfilled} evals={st['evaluations']:.3e}", flush=True)
    except Exception as e:
        print(f"{tag} M={m} cap={cap}: ERROR {e}", flush=True)
        break

import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from cases import make_case
from synference_amd.engine import HipFlow
ospec, spec, flat, theta, x = make_case("maf_cfg1", B=2000, spread=0.2)
f = HipFlow(spec, "cuda:0"); f.set_params(torch.as_tensor(flat))
lo = (np.asarray(ospec.theta_mean) - 1.0 * np.asarray(ospec.theta_std)).astype(np.float32)
hi = (np.asarray(ospec.theta_mean) + 1.0 * np.asarray(ospec.theta_std)).astype(np.float32)
X = torch.as_tensor(x).cuda()
for _ in range(2): acc = f.acceptance(X, 10000, lo, hi, seed=3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): acc = f.acceptance(X, 10000, lo, hi, seed=3)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(os.environ.get("SF_FIND16S", "1"), "acceptance of 2000 contexts x 10000 draws: %.2f ms = %.2f G evals/s; mean acc %.4f" % (dt * 1e3, 2e7 / dt / 1e9, float(acc.float().mean()) / 10000))

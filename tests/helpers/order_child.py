"""Child process of test_gpu_sampler_stress.py::test_draws_do_not_depend_on_the_dense_order: samples one catalogue with
the dense-list order this process's environment selects (SF_INTERLEAVE is read once per process) and prints a digest of
the result: sha256 of the bytes of the output and of the attempt counts."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cases import make_case  # noqa: E402
from oracle import posterior as OP  # noqa: E402  (only to place the box: the checker, not the thing measured)
from synference_amd.engine import HipFlow  # noqa: E402

name, M, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
qlo = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1     # the box: central (1 - 2 qlo) quantile range per dimension
ospec, spec, flat, theta, x = make_case(name, B=M, spread=0.2)
free, _ = OP.sample(ospec, torch.as_tensor(flat), x[:16], 200 if qlo <= 0.1 else 2000, 5, dtype=torch.float32)
free = free.reshape(-1, ospec.D)
lo = np.quantile(free, qlo, axis=0).astype(np.float32)
hi = np.quantile(free, 1.0 - qlo, axis=0).astype(np.float32)
f = HipFlow(spec, "cuda:0")
f.set_params(torch.as_tensor(flat))
got, nd = f.sample(x, S, lo, hi, seed=11, return_counts=True)
h = hashlib.sha256()
h.update(got.cpu().numpy().tobytes())
h.update(nd.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest(), int(f.last_unfilled), float(torch.nan_to_num(got).double().sum()), int(f.last_sample_stats["rounds"]))

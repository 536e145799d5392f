"""loss_grad's per-row losses of the lampe flow against -log_prob (another kernel) at batch sizes around chunk boundaries: prints the rows
that disagree -- none since the barrier behind the loss (see the note above k_ar_train in csrc/sf_nsfar.hip)."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from cases import make_case
from synference_amd.engine import HipFlow
ospec, spec, flat, theta, x = make_case("nsfar_cfg1", B=40000)
f = HipFlow(spec, "cuda:0"); fl = torch.as_tensor(flat)
T, X = torch.as_tensor(theta).cuda(), torch.as_tensor(x).cuda()
f.set_params(fl.cuda())
ref = -f.log_prob(T, X)
for B in (20000, 20032, 40000, 19968):
    for rep in range(2):
        l, g = f.loss_grad(fl, T[:B], X[:B], 1.0 / 40000)
        d = (l - ref[:B]).abs()
        bad = (d > 1e-3).nonzero().flatten()
        print("B", B, "rep", rep, "max |loss - (-log_prob)|", float(d.max()), "rows off", bad.numel(), bad[:8].tolist(), bad[-4:].tolist())

"""One-off randomized sweep of the 32-row MAF paths (SF_MAF16=0 forces them) against the oracle."""
import os, sys
os.environ["SF_MAF16"] = "0"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import flows as OF
from synference_amd.spec import FlowSpec
from synference_amd.engine import HipFlow
rng = np.random.default_rng(78)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
worst = 0.0; ninc = 0; nspan = 0
for i in range(n):
    D = int(rng.integers(2, 17)); H = int(rng.integers(2, 129)); NB = int(rng.integers(1, 3)); T = int(rng.integers(1, 3))
    C = int(rng.choice([1, 5, 16, 33])); seed = 2000 + i
    perms = OF.random_perms(D, T, seed)
    st = dict(theta_mean=rng.normal(size=D).astype(np.float32), theta_std=rng.uniform(0.5, 2, size=D).astype(np.float32),
              x_mean=rng.normal(size=C).astype(np.float32), x_std=rng.uniform(0.5, 2, size=C).astype(np.float32))
    ospec = OF.FlowSpec(kind="maf", D=D, C=C, H=H, T=T, NB=NB, perms=perms, **{k: v.astype(np.float64) for k, v in st.items()})
    spec = FlowSpec(kind="maf", D=D, C=C, H=H, T=T, NB=NB, perms=perms, **st)
    flat = OF.init_params(ospec, seed + 1)
    flat = (flat + 0.4 * rng.normal(size=flat.shape) * np.abs(flat).mean()).astype(np.float32)
    B = 37
    theta = (rng.normal(size=(B, D)) * st["theta_std"] + st["theta_mean"]).astype(np.float32)
    x = (rng.normal(size=(B, C)) * st["x_std"] + st["x_mean"]).astype(np.float32)
    z = rng.normal(size=(B, D)).astype(np.float32)
    f = HipFlow(spec, "cuda:0"); f.set_params(torch.as_tensor(flat))
    d = f.describe(); ninc += d["inc_ok"]; nspan += int(d["g_lo"] != d["g_tile"])
    th, ld = f.inverse(z, x)
    pt = torch.as_tensor(flat, dtype=torch.float64)
    th_ref, ld_ref = OF.inverse_transform(ospec, pt, torch.as_tensor(z).double(), torch.as_tensor(x).double())
    scale = np.maximum(np.abs(th_ref.numpy()), st["theta_std"])
    e = float(np.abs((th.cpu().double().numpy() - th_ref.numpy()) / scale).max()); e2 = float(np.abs(ld.cpu().double().numpy() - ld_ref.numpy()).max())
    lp = f.log_prob(theta, x).cpu().double().numpy()
    e3 = float(np.abs(lp - OF.log_prob(ospec, pt, torch.as_tensor(theta).double(), torch.as_tensor(x).double()).numpy()).max())
    worst = max(worst, e, e2 / max(1.0, D / 4), e3 / max(1.0, D / 4))
    if e > 5e-4 or e2 > 5e-4 * max(1.0, D / 4) or e3 > 1e-4 * max(1.0, D / 4):
        print("BAD", dict(D=D, H=H, NB=NB, T=T, C=C, inc=d["inc_ok"]), e, e2, e3)
print(f"{n} shapes, {ninc} incremental, {nspan} with straddling groups, worst error {worst:.2e}")

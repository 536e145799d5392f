"""Training-kernel timings: cooperative 16-row kernel (SF_TRAINC=1, default) vs the one-wave-per-tile kernel (SF_TRAINC=0).
Run once per setting (the switch is read once per process): prints HIP-event kernel ms and wall ms per loss_grad."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd.spec import FlowSpec, init_params, random_perms
from synference_amd.engine import HipFlow
g = torch.Generator().manual_seed(0)
KIND = os.environ.get("SF_PROBE_KIND", "maf")   # maf: BASELINE cfg1 shape; nsf: cfg3 shape
D, C = (5, 10) if KIND == "maf" else (8, 20)
D, C = int(os.environ.get("SF_PROBE_D", D)), int(os.environ.get("SF_PROBE_C", C))
TM = int(os.environ.get("SF_PROBE_T", "5"))
spec = (FlowSpec(kind="maf", D=D, C=C, H=int(os.environ.get("SF_PROBE_H", "50")), T=TM, K=10, perms=random_perms(D, TM, g)) if KIND == "maf" else
        FlowSpec(kind="nsf", D=D, C=C, H=int(os.environ.get("SF_PROBE_H", "50")), T=int(os.environ.get("SF_PROBE_T", "5")),
                 K=int(os.environ.get("SF_PROBE_K", "8"))))
F_LP = 40030 * TM / 5 if KIND == "maf" else 178640
f = HipFlow(spec); flat = init_params(spec, g).cuda(); grad = torch.empty_like(flat)
f.set_profiling(True)
for B in [int(b) for b in os.environ.get("SF_PROBE_BS", "64,2048,16384,131072").split(",")]:
    th = torch.randn(B, D, device="cuda"); x = torch.randn(B, C, device="cuda")
    for _ in range(3): f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)
    torch.cuda.synchronize()
    ks = []
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        f.loss_grad(flat, th, x, 1.0 / B, grad_out=grad)
        ks.append(f.train_kernel_ms())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    k = float(np.median(ks))
    print(f"{KIND} SF_TRAINC={os.environ.get('SF_TRAINC','1')} path={f.train_path(B)} B={B}: kernel {k*1e3:.1f} us (min {min(ks)*1e3:.1f}), wall {dt*1e3:.3f} ms, "
          f"{3*F_LP*B/(k*1e-3)/1e12:.2f} TFLOP/s = {3*F_LP*B/(k*1e-3)/157.3e12:.3f} of fp32 peak", flush=True)

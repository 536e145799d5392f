"""End-to-end host-visible timings of the SBI_Fitter surface (diagnostics)."""
import os, sys, time
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd import SBI_Fitter
from synference_amd.synthetic import make_catalogue

kind = sys.argv[1] if len(sys.argv) > 1 else "maf"
D, C = (5, 10) if kind == "maf" else (8, 20)
x, th, names = make_catalogue(20000, C, D, seed=1234)
filt = [f"F{i}" for i in range(C)]
f = SBI_Fitter("m", names, filt, feature_array=x, parameter_array=th)
t0 = time.perf_counter()
post, stats = f.run_single_sbi(backend="hip", engine="NPE", model_type=kind, hidden_features=50, num_transforms=5,
                               training_batch_size=64, stop_after_epochs=5, max_num_epochs=8, plot=False, save_model=False,
                               additional_model_args={"num_bins": 8} if kind == "nsf" else {})
t1 = time.perf_counter()
print(f"run_single_sbi (batch 64, {stats[0]['epochs_trained']} epochs, {int(0.8*0.8*20000)} train rows): {t1 - t0:.2f} s")
xo, _, _ = make_catalogue(20000, C, D, seed=99)
for n in (1000, 20000):
    f.sample_posterior(xo[:64], num_samples=100)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = f.sample_posterior(xo[:n], num_samples=1000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"sample_posterior N={n} S=1000 -> {s.shape} {s.dtype}: {dt*1e3:.1f} ms = {dt/n*1e6:.2f} us/object")
    df = pd.DataFrame(xo[:n], columns=filt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tab = f.fit_catalogue(df, num_samples=1000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"fit_catalogue   N={n} S=1000: {dt*1e3:.1f} ms = {dt/n*1e6:.2f} us/object")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lp = f.log_prob(xo[:n], th[:n])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"log_prob        N={n}: {dt*1e3:.1f} ms")

"""Where the time of SBI_Fitter.sample_posterior goes (bench `api` leg): engine call, D2H, float64 widening, page faults.
Prints one line per stage / setting; run on the GPU box."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synference_amd import hostio

N, S, D = 2000, 1000, 5
dev = torch.device("cuda:0")
x = torch.randn(N, S, D, device=dev)
torch.cuda.synchronize()


def t(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print(f"usable cores {hostio.usable_cores()}", flush=True)
print(f"np.empty + touch (1 thread)          {t(lambda: np.empty((N, S, D)).fill(0.0)):.2f} ms")
pin = torch.empty((N, S, D), dtype=torch.float32, pin_memory=True)
print(f"D2H 40 MB into pinned (one copy)      {t(lambda: (pin.copy_(x, non_blocking=True), torch.cuda.synchronize())):.2f} ms")
out = np.empty((N, S, D))
print(f"widen pinned f32 -> f64 (1 thread)    {t(lambda: np.copyto(out, pin.numpy(), casting='same_kind')):.2f} ms")
print(f".double().cpu().numpy() (old path)    {t(lambda: x.double().cpu().numpy()):.2f} ms")
for chunk in (1.0, 2.0, 4.0, 8.0, 16.0):
    for workers in (4, 8, 15):
        for nb in (3, 6):
            ms_new = t(lambda: hostio.to_host_f64(x, chunk_mb=chunk, n_buf=nb, workers=workers))
            ms_reuse = t(lambda: hostio.to_host_f64(x, out=out, chunk_mb=chunk, n_buf=nb, workers=workers))
            print(f"to_host_f64 chunk {chunk:4.1f} MB workers {workers:2d} bufs {nb}: fresh array {ms_new:.2f} ms, reused array {ms_reuse:.2f} ms", flush=True)

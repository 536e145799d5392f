"""Reads the iteration trace of the SF_Q_STATS build (SF_Q_TRACE=file): utilisation of the persistent sampler over time."""
import sys
import numpy as np
ipw = int(sys.argv[2]) if len(sys.argv) > 2 else 128
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 1024          # workgroups of the launch: [grid][2048 * 256 / grid][4]
step = float(sys.argv[4]) if len(sys.argv) > 4 else 100.0        # report window, us
tr = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(grid, (2048 * 256) // grid, 4)
used = tr[:, :, 1] > 0
wg = used.any(axis=1)
t0 = tr[:, :, 0][used].min()
start = (tr[:, :, 0].astype(np.int64) - t0) * 0.01          # us
end = (tr[:, :, 3].astype(np.int64) - t0) * 0.01
ent = tr[:, :, 1].astype(np.int64)
lg = (tr[:, :, 2] & 0xff).astype(np.int64)
tail = (tr[:, :, 2] >> 8) & 1
items = ent << lg
print(f"workgroups {wg.sum()}, iterations {used.sum()}, last end {end[used].max():.0f} us, items {items[used].sum()}")
dense_end = start[used & (tail == 0)].max()
print(f"last dense iteration starts at {dense_end:.0f} us; first tail iteration at {start[used & (tail == 1)].min():.0f} us")
edges = np.arange(0, end[used].max() + step, step)
print(" window_us  active_wgs  mean_items/iter  iter_us  lgA_hist")
for a, b in zip(edges[:-1], edges[1:]):
    sel = used & (start >= a) & (start < b)
    if not sel.any():
        continue
    # workgroups inside an iteration at the middle of the window
    mid = (a + b) / 2
    active = (used & (start <= mid) & (end > mid)).any(axis=1).sum()
    dur = (end - start)[sel]
    h = np.bincount(lg[sel], minlength=7)
    print(f"{a:8.0f}  {active:8d}  {items[sel].mean():10.1f}/{ipw}  {np.median(dur):8.1f}  {h.tolist()}")

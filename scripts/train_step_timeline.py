"""From a rocprofv3 --kernel-trace run of bench.py: the kernels of one steady-state training step at the bench batch, with
their durations and the idle gaps between them (what a step's wall time is made of)."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
grid = sys.argv[2] if len(sys.argv) > 2 else '131072'
idx = [i for i, r in enumerate(rows) if 'k_maf_trainc' in r['Kernel_Name'] and r['Grid_Size_X'] == grid]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
for r in rows[a:b + 1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(st - t0) / 1e3:8.1f} us  gap {gap:6.1f}  dur {(en - st) / 1e3:7.1f} us  {r['Kernel_Name'][:60]}")
    prev_end = en

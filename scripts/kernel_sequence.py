import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find last persistent launch, print everything from the one before it
idx = [i for i, r in enumerate(rows) if 'k_sample_persist' in r['Kernel_Name'] or 'k_maf_samp16' in r['Kernel_Name']]
a = idx[-2] if len(idx) >= 2 else 0
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a: idx[-1] + 1]:
    n = r['Kernel_Name'][:70]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  grid {r['Grid_Size_X']:>9s}  {n}")

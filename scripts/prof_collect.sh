#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + three PMC passes of the default bench command,
# keeps only small summaries under gpurun_out/prof/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof; RAW=/tmp/sfprof; rm -rf $RAW; mkdir -p $OUT $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $RAW/stats.err
cp $RAW/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
python3 - "$RAW" "$OUT" <<'PY'
import csv, glob, sys, statistics
raw, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(raw + '/stats/*/*_kernel_trace.csv')[0])))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
isk = lambda r: 'k_inverse' in r['Kernel_Name'] or 'k_maf_inv16' in r['Kernel_Name']
big = [r for r in rows if isk(r) and int(r['Grid_Size_X']) > 1000000]
small = [r for r in rows if isk(r) and int(r['Grid_Size_X']) <= 1000000]
ctab = [r for r in rows if 'ctab' in r['Kernel_Name']]
with open(out + '/round0_kernel.txt', 'w') as f:
    f.write(f"kernel {big[0]['Kernel_Name']}\n")
    f.write(f"dense round-0 launches n={len(big)} avg_us={statistics.mean(map(dur, big)):.1f} min_us={min(map(dur, big)):.1f} max_us={max(map(dur, big)):.1f}\n")
    f.write(f"VGPR={big[0]['VGPR_Count']} AGPR={big[0]['Accum_VGPR_Count']} SGPR={big[0]['SGPR_Count']} LDS={big[0]['LDS_Block_Size']} scratch={big[0]['Scratch_Size']} grid={big[0]['Grid_Size_X']} wg={big[0]['Workgroup_Size_X']}\n")
    f.write(f"retry-round launches n={len(small)} median_us={statistics.median(map(dur, small)):.1f}\n")
    if ctab: f.write(f"context-table launches n={len(ctab)} median_us={statistics.median(map(dur, ctab)):.1f} ({ctab[0]['Kernel_Name']})\n")
PY
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $RAW/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $RAW/pmc_$tag.err
  python3 - "$RAW/pmc_$tag" "$OUT/pmc_${tag}_round0.csv" <<'PY'
import csv, glob, sys
rows = [r for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]))
        if ('k_inverse' in r['Kernel_Name'] or 'k_maf_inv16' in r['Kernel_Name']) and int(r['Grid_Size']) > 1000000]
w = csv.DictWriter(open(sys.argv[2], 'w'), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
PY
done
python3 bench.py > $OUT/bench_default.json 2> /dev/null
ls -la $OUT

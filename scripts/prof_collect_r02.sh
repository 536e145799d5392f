#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + three PMC passes of the default bench command (sampler kernel
# k_maf_samp16 and training kernel k_maf_train); keeps only small summaries under gpurun_out/prof_r02/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r02; RAW=/tmp/sfprof; rm -rf $RAW; mkdir -p $OUT $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $RAW/stats.err
cp $RAW/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv
python3 - "$RAW" "$OUT" <<'PY'
import csv, glob, sys, statistics
raw, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(raw + '/stats/*/*_kernel_trace.csv')[0])))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
def block(f, tag, sel):
    rs = [r for r in rows if sel(r)]
    if not rs: return
    d = list(map(dur, rs))
    f.write(f"{tag}: kernel {rs[0]['Kernel_Name']}\n")
    f.write(f"  launches n={len(rs)} avg_us={statistics.mean(d):.1f} median_us={statistics.median(d):.1f} min_us={min(d):.1f} max_us={max(d):.1f}\n")
    f.write(f"  VGPR={rs[0]['VGPR_Count']} AGPR={rs[0]['Accum_VGPR_Count']} SGPR={rs[0]['SGPR_Count']} LDS={rs[0]['LDS_Block_Size']} scratch={rs[0]['Scratch_Size']} grid={rs[0]['Grid_Size_X']} wg={rs[0]['Workgroup_Size_X']}\n")
with open(out + '/kernels.txt', 'w') as f:
    # the timed sampling steps are the launches over the full 2e6-slot catalogue; the 20 timed + 3 warm-up ones come first
    block(f, "sampler (persistent, one launch per bench step)", lambda r: 'k_maf_samp16' in r['Kernel_Name'])
    block(f, "context table", lambda r: 'ctab' in r['Kernel_Name'])
    tiles = lambda r: int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])   # one workgroup (1 + NC waves) per 32-row tile
    block(f, "train fwd+bwd at batch 16384", lambda r: 'k_maf_train' in r['Kernel_Name'] and tiles(r) == 512)
    block(f, "train fwd+bwd at batch 2048 (warm-up fit)", lambda r: 'k_maf_train' in r['Kernel_Name'] and tiles(r) == 64)
    block(f, "train fwd+bwd at batch 64", lambda r: 'k_maf_train' in r['Kernel_Name'] and tiles(r) == 2)
    block(f, "log_prob 200k rows", lambda r: 'k_logprob' in r['Kernel_Name'])
    block(f, "clip + Adam", lambda r: 'k_adam' in r['Kernel_Name'])
    block(f, "train prep (re-tile + zero)", lambda r: 'k_train_prep' in r['Kernel_Name'])
    block(f, "gradient gather", lambda r: 'k_grad_gather' in r['Kernel_Name'])
PY
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $RAW/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $RAW/pmc_$tag.err
  python3 - "$RAW/pmc_$tag" "$OUT/pmc_${tag}" <<'PY'
import csv, glob, sys
allr = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0])))
for tag, sel in (("sampler", lambda r: 'k_maf_samp16' in r['Kernel_Name']),
                 ("train16384", lambda r: 'k_maf_train' in r['Kernel_Name'] and int(r['Grid_Size']) // int(r['Workgroup_Size']) == 512)):
    rows = [r for r in allr if sel(r)]
    if rows:
        w = csv.DictWriter(open(f"{sys.argv[2]}_{tag}.csv", 'w'), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
PY
done
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
ls -la $OUT

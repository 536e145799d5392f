#!/bin/bash
# Runs on the GPU box (gpurun): everything behind profiles/rNN_*.      usage: bash scripts/prof_collect.sh r05
#   1. rocprofv3 --kernel-trace --stats of the default bench command          -> kernel_stats.csv, kernels.txt
#   2. three separate --pmc passes of the default bench (FETCH_SIZE | WRITE_SIZE | SQ_*)  -> per-kernel counter rows
#   3. the same for the NSF workload of BASELINE configs[2] (bench.py --workload nsf_cfg3)
#   4. un-profiled bench lines of both workloads
#   5. (round 5) kernel trace + PMC passes of bench.py --workload nsfar_cfg2 (the lampe backend's flow: k_ar_samp16)
#   6. (round 4) bench.py --workload nsf_prod: the reference's production NSF (T = 15, H = 69, K = 10);
#      bench.py --workload nsfar_cfg2: the lampe backend's autoregressive NSF on the cfg2 mock (with its CPU baseline)
# Only small summaries are kept (gpurun_out/prof_rNN/); scripts/make_pmc_summary.py rNN turns them into
# profiles/rNN_pmc_summary.json, which bench.py reads `traffic` / `issue_busy` from.
set -e
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; RAW=/tmp/sfprof_$TAG; rm -rf $RAW; mkdir -p $OUT $RAW
say() { echo "[$(date +%H:%M:%S)] $*" | tee -a $OUT/progress.log; }

trace() {  # tag, bench args
  rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats_$1 -- python3 bench.py $2 --no-cpu-baseline --skip-large-catalogue --skip-nsf-leg --skip-lampe-leg --skip-per-object --repeats 1 > $OUT/bench_under_rocprof_$1.json 2> $RAW/stats_$1.err
  cp $RAW/stats_$1/*/*_kernel_stats.csv $OUT/kernel_stats_$1.csv
  python3 - "$RAW/stats_$1" "$OUT/kernels_$1.txt" <<'PY'
import csv, glob, sys, statistics, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0])))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
groups = collections.defaultdict(list)
for r in rows:
    groups[(r['Kernel_Name'], r['Grid_Size_X'], r['Workgroup_Size_X'])].append(r)
with open(sys.argv[2], 'w') as f:
    for key, rs in sorted(groups.items(), key=lambda kv: -sum(map(dur, kv[1]))):
        d = list(map(dur, rs))
        if sum(d) < 50: continue
        f.write(f"{key[0]}\n  grid={key[1]} wg={key[2]} launches n={len(rs)} avg_us={statistics.mean(d):.1f} median_us={statistics.median(d):.1f} "
                f"min_us={min(d):.1f} max_us={max(d):.1f} total_ms={sum(d)/1e3:.2f}\n"
                f"  VGPR={rs[0]['VGPR_Count']} AGPR={rs[0]['Accum_VGPR_Count']} SGPR={rs[0]['SGPR_Count']} LDS={rs[0]['LDS_Block_Size']} scratch={rs[0]['Scratch_Size']}\n")
PY
}

pmc() {  # tag, bench args, kernel regex
  for pass in "FETCH_SIZE" "WRITE_SIZE" \
              "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY" \
              "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM"; do
    t=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $RAW/pmc_$1_$t -- python3 bench.py $2 --no-cpu-baseline --skip-large-catalogue --skip-nsf-leg --skip-lampe-leg --skip-per-object --skip-dp --repeats 1 > /dev/null 2> $RAW/pmc_$1_$t.err
    say "pmc $1 $t done"
  done
  python3 - "$RAW" "$1" "$3" "$OUT/pmc_$1.csv" <<'PY'
# per (kernel, grid, workgroup size, counter): mean counter value, launches, mean duration -- a few hundred rows
import csv, glob, sys, re, collections
raw, tag, pat, out = sys.argv[1], sys.argv[2], re.compile(sys.argv[3]), sys.argv[4]
acc = collections.defaultdict(lambda: [0.0, 0, 0.0]); meta = {}
for f in sorted(glob.glob(f"{raw}/pmc_{tag}_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if not pat.search(r['Kernel_Name']): continue
        key = (r['Kernel_Name'], r['Grid_Size'], r['Workgroup_Size'], r['Counter_Name'])
        a = acc[key]
        a[0] += float(r['Counter_Value']); a[1] += 1; a[2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        meta[key[:3]] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
with open(out, 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size',
                'Scratch_Size', 'Counter_Name', 'Counter_Mean', 'Launches', 'Mean_us'])
    for key in sorted(acc):
        a = acc[key]
        w.writerow(list(key[:3]) + list(meta[key[:3]]) + [key[3], a[0] / a[1], a[1], a[2] / a[1]])
print(len(acc), "aggregated counter rows ->", out)
PY
}

say "kernel trace, default bench"
trace maf ""
say "PMC passes, default bench"
pmc maf "--steps 3 --warmup 1 --skip-throughput-regime --skip-api" "k_maf_samp16|k_maf_find16s|k_maf_ctab16|k_maf_trainc|k_gather_c|k_train_prep|k_adam|k_logprob"
say "kernel trace, nsf_cfg3"
trace nsf "--workload nsf_cfg3 --steps 3 --warmup 1"
say "PMC passes, nsf_cfg3"
pmc nsf "--workload nsf_cfg3 --steps 2 --warmup 1 --skip-throughput-regime --skip-api" "k_sample_persist|k_logprob|k_nsf_train|k_gather_c|k_train_prep"
say "PMC passes, nsf_cfg3 training with the gradient replicas (SF_GRAD_ACC=atomic)"
export SF_GRAD_ACC=atomic
pmc nsfatomic "--workload nsf_cfg3 --steps 2 --warmup 1 --skip-throughput-regime --skip-api --fit-steps 50" "k_nsf_train|k_gather_c"
unset SF_GRAD_ACC
say "kernel trace, nsfar_cfg2 (the lampe backend's flow)"
trace nsfar "--workload nsfar_cfg2 --steps 3 --warmup 1"
say "PMC passes, nsfar_cfg2"
pmc nsfar "--workload nsfar_cfg2 --steps 3 --warmup 1 --skip-throughput-regime --skip-api" "k_ar_samp16|k_ar_find16|k_ar_resolve16|k_ar_train|k_ar_logprob"
say "un-profiled bench lines"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
say "default bench done"
python3 bench.py --workload nsf_cfg3 --no-cpu-baseline > $OUT/bench_nsf_cfg3.json 2> $OUT/bench_nsf_cfg3.err
python3 bench.py --workload nsf_prod --no-cpu-baseline --skip-large-catalogue > $OUT/bench_nsf_prod.json 2> $OUT/bench_nsf_prod.err
say "nsf_prod bench done"
python3 bench.py --workload nsfar_cfg2 --skip-large-catalogue > $OUT/bench_nsfar_cfg2.json 2> $OUT/bench_nsfar_cfg2.err
say "nsfar_cfg2 bench done"
say "all done"
ls -la $OUT

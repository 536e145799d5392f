#!/bin/bash
# On the GPU box: SQ / TCC counter passes over one bench.py workload; per-kernel averages of the kernels whose name
# matches a pattern.  usage: pmc_bench.sh TAG "bench args" "kernel-name regex"
set -e
TAG=$1; ARGS=$2; PAT=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/pmcb_$TAG; rm -rf $RAW; mkdir -p $RAW gpurun_out
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY" \
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $RAW/p$i -- python3 bench.py $ARGS --no-cpu-baseline > $RAW/p$i.out 2> $RAW/p$i.err || { tail -5 $RAW/p$i.err; exit 1; }
  echo "pass $i done" >> gpurun_out/pmc_bench_$TAG.progress
done
python3 - "$RAW" "$TAG" "$PAT" <<'PY' | tee gpurun_out/pmc_bench_$TAG.txt
import csv, glob, sys, collections, re
raw, tag, pat = sys.argv[1], sys.argv[2], re.compile(sys.argv[3])
acc = collections.defaultdict(lambda: collections.defaultdict(list)); durs = collections.defaultdict(list); meta = {}
for f in glob.glob(raw + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not pat.search(k): continue
        key = (k, r['Grid_Size'], r['Workgroup_Size'])
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        durs[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        meta[key] = (r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('LDS_Block_Size'), r.get('Scratch_Size'))
for key in sorted(acc, key=lambda k: -sum(durs[k])):
    print(f"== {tag} | {key[0]} grid={key[1]} wg={key[2]} vgpr/agpr/lds/scratch={meta[key]}")
    print(f"{'kernel_us_under_pmc':32s} {sum(durs[key])/len(durs[key]):16.1f}  n={len(durs[key])}")
    for c in sorted(acc[key]): print(f"{c:32s} {sum(acc[key][c])/len(acc[key][c]):16.0f}  n={len(acc[key][c])}")
PY

#!/bin/bash
# On the GPU box: PMC passes over scripts/time_train_kernel.py, rows of the k_*_train kernel.
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RAW=/tmp/pmctrain_$TAG; rm -rf $RAW; mkdir -p $RAW gpurun_out
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_WR" \
            "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $RAW/p$i -- python3 scripts/time_train_kernel.py > $RAW/p$i.out 2> $RAW/p$i.err || { tail -5 $RAW/p$i.err; exit 1; }
done
python3 - "$RAW" "$TAG" <<'PY' | tee gpurun_out/pmc_train_$1.txt
import csv, glob, sys, collections
raw, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list); name = None; durs = []
for f in glob.glob(raw + '/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if '_train' in r['Kernel_Name'] and 'prep' not in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])); name = r['Kernel_Name']
            durs.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
            regs = (r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('LDS_Block_Size'), r.get('Scratch_Size'), r['Grid_Size'], r['Workgroup_Size'])
print(tag, name, 'vgpr/agpr/lds/scratch/grid/wg', regs)
print(f"{'kernel_us_under_pmc':32s} {sum(durs)/len(durs):16.1f}  n={len(durs)}")
for k in sorted(acc): print(f"{k:32s} {sum(acc[k])/len(acc[k]):16.0f}  n={len(acc[k])}")
PY

#!/bin/bash
# quick timing of the lampe NSF bench workload on the GPU box: bash scripts/ab_nsfar.sh [n]
for i in $(seq ${1:-2}); do
  python bench.py --workload nsfar_cfg2 --skip-large-catalogue --no-cpu-baseline --skip-throughput-regime --skip-api --skip-per-object 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; t=d.get('roofline_train') or {}
print('value %.4g ms_per_step %.3f sampler_launch_ms %.3f frac %.4f | train launch_ms %.4f frac %.4f' % (d['value'], d['ms_per_step'], r['launch_ms'], r['frac'], t.get('launch_ms', 0), t.get('frac', 0)))"
done

#!/bin/bash
# usage: ab_nsf.sh n "ENV_A" "ENV_B" ... : alternating runs of the NSF workload (BASELINE configs[2])
n=$1; shift
for i in $(seq 1 $n); do for v in "$@"; do env $v timeout -k 10 400 python bench.py --workload nsf_cfg3 --no-cpu-baseline --skip-throughput-regime --skip-large-catalogue --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$v', 'ms_per_step', round(d['ms_per_step'], 2), 'kernel_ms', round(d['roofline'].get('launch_ms'), 2), 'frac', round(d['roofline']['frac'], 4), 'value', round(d['value'] / 1e6, 1))"; done; done

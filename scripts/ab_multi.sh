#!/bin/bash
# usage: ab_multi.sh n "ENV_A" "ENV_B" ...  -> n rounds of alternating bench runs, prints the sampler launch time of each
n=$1; shift
for i in $(seq 1 $n); do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --skip-throughput-regime --skip-large-catalogue --skip-nsf-leg --skip-lampe-leg --skip-api --skip-per-object --skip-dp --repeats 1 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$v', 'ms_per_step', round(d['ms_per_step'], 3), 'kernel_ms', round(d['roofline'].get('launch_ms'), 3), 'frac', round(d['roofline']['frac'], 4), 'evals', d['config'].get('evals_per_step', d['roofline'].get('note', '')[:0]))"
  done
done

for i in 1 2 3; do timeout -k 10 300 python bench.py --no-cpu-baseline --skip-large-catalogue --skip-throughput-regime --steps 3 --warmup 1 --train-steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d['train']; print('train ms/step', round(t['ms_per_step'], 4), 'Mpairs/s', round(t['value'] / 1e6, 1), 'batch64 ms/step', round(t['batch64_ms_per_step'], 4), 'kernel', round(d['roofline_train']['launch_ms'], 4))"; done

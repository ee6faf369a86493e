run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline > /tmp/o.json 2>/dev/null; python3 -c "
import json,os; d=json.load(open('/tmp/o.json')); r=d['roofline']; print('$1', 'frame', round(d['ms_per_step'],4), 'kernel', round(r['kernel_ms'],4))"; }
LS_PROJECT_DEBUG=0 run full
LS_PROJECT_DEBUG=1 run loads_only
LS_PROJECT_DEBUG=2 run loads_footprint
LS_PROJECT_DEBUG=3 run all_but_tests
LS_PROJECT_EXTRA_LDS=20000 run occupancy_4blocks
LS_PROJECT_EXTRA_LDS=60000 run occupancy_2blocks

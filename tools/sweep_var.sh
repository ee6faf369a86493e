#!/bin/bash
# bench the library variants under build/var/ (built by hand with -D switches), default library first
W=${W:-syn128x1m}
run() { python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > /tmp/o.json 2>/dev/null; python3 -c "
import json,os; d=json.load(open('/tmp/o.json')); r=d['roofline']; print('$1', 'frame', round(d['ms_per_step'],4), 'kernel', round(r['kernel_ms'],4), 'tests', r.get('candidate_tests_per_launch'))"; }
run default
for f in build/var/lib_*.so; do LS_LIB_PATH=$PWD/$f run $(basename $f); done

#!/bin/bash
# NEEDS an experimental build: the LS_PROJECT_* / LS_TRACE_* knobs are only read by a library built with -DLS_EXPERIMENTAL
# (make -C lidarshooter_amd/csrc clean all EXPERIMENTAL=1, or tools/exp_build.sh + LS_LIB_PATH); the shipped library ignores them.
# k_project ablation: LS_PROJECT_DEBUG stops the kernel after a phase (1 = vertex loads, 2 = stage 1,
# 3 = stage 2 footprints); prints frame and kernel time for each.
W=${W:-syn128x1m}
run() { python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > /tmp/o.json 2>/dev/null; python3 -c "
import json,os; d=json.load(open('/tmp/o.json')); r=d['roofline']; print('$1', 'frame', round(d['ms_per_step'],4), 'kernel', round(r['kernel_ms'],4), 'tests', r.get('candidate_tests_per_launch'))"; }
LS_PROJECT_DEBUG=0 run full
LS_PROJECT_DEBUG=1 run loads_only
LS_PROJECT_DEBUG=2 run stage1
LS_PROJECT_DEBUG=3 run stage2_footprints

#!/bin/bash
# What the device-wide wait at a window's end costs the driver's own command (`bench.py --steps 20`): the runtime's wait policy
# A/B -- as shipped / HSA_ENABLE_INTERRUPT=0 (signal waits poll) / ROC_ACTIVE_WAIT_TIMEOUT (us of polling before the blocked wait).
# usage (GPU box, repo root): bash tools/sync_wait_ab.sh [repeats]
N=${1:-3}
OUT=gpurun_out/sync_wait_ab
mkdir -p $OUT
run() {  # label, env...
  local label=$1; shift
  for i in $(seq $N); do
    env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --no-also 2>> $OUT/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
t = d['timing']['window_ms_min_median_max']
print('$label: %.2f us per frame (windows of 20: min %.2f median %.2f max %.2f), windows of 1000 %.2f, host %.2f us' % (d['ms_per_step']*1e3, t[0]/20*1e3, t[1]/20*1e3, t[2]/20*1e3, d['ms_per_step_windows_of_1000']*1e3, d['host_enqueue_ms_per_step']*1e3))"
  done
}
run "as shipped           " LS_NOP=1
run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
run "ROC_ACTIVE_WAIT_TIMEOUT=200" ROC_ACTIVE_WAIT_TIMEOUT=200
run "ROC_ACTIVE_WAIT_TIMEOUT=100000" ROC_ACTIVE_WAIT_TIMEOUT=100000
run "as shipped (again)   " LS_NOP=1

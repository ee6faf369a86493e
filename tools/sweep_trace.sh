# NEEDS an experimental build: the LS_PROJECT_* / LS_TRACE_* knobs are only read by a library built with -DLS_EXPERIMENTAL
# (make -C lidarshooter_amd/csrc clean all EXPERIMENTAL=1, or tools/exp_build.sh + LS_LIB_PATH); the shipped library ignores them.
run() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline > /tmp/o.json 2>/dev/null; python3 -c "
import json,os; d=json.load(open('/tmp/o.json')); r=d['roofline']; print('$1', 'frame', round(d['ms_per_step'],4), 'trace', round(r['kernel_ms'],4), 'maxtrips', r['wave_trips_max'], 'trace_only', round(d.get('trace_only_ms',0),4))"; }
LS_TRACE_LOAD_MODE=0 run plain
LS_TRACE_LOAD_MODE=1 run nt
LS_TRACE_LOAD_MODE=1 LS_TRACE_BLOCKS_PER_CU=4 run nt_b4
LS_TRACE_LOAD_MODE=1 LS_TRACE_BLOCKS_PER_CU=8 run nt_b8
LS_TRACE_LOAD_MODE=1 LS_TRACE_CHAN_MUL=1 run nt_seq

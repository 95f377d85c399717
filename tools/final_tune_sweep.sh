#!/bin/bash
# last tuning pass on the final kernels: round width, probe cadence, probe margin, big-variant grid (each line one default bench without the CPU baseline)
run() { echo "== $*"; env "$@" python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s nodes %d roofline %.3f group ms %.2f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), c['bnb_nodes'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms', 0)))
"; }
run MIQP_X=0
run MIQP_X=0
run MIQP_NPR=24
run MIQP_NPR=48
run MIQP_PROBE_EVERY=2
run MIQP_PROBE_MARGIN=0.5
run MIQP_BIG_GRID=640

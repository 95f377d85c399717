#!/bin/bash
# the probe experiments again, now that only rounding probes carry the probe mark (before, first children did too)
run() { echo "== $*"; env "$@" python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s nodes %d roofline %.3f launch ms %.2f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), c['bnb_nodes'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms', 0)))
"; }
run MIQP_PROBE_MARGIN=0.25
run MIQP_PROBE_MARGIN=0.25 MIQP_PROBE_EVERY=2
run MIQP_PROBE_MARGIN=0.25 MIQP_PROBE_ITCAP=24
run MIQP_PROBE_MARGIN=0.25 MIQP_PROBE_ITCAP=16
run MIQP_PROBE_MARGIN=0
run MIQP_PROBE_MARGIN=0 MIQP_OC_BIG=0

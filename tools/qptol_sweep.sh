#!/bin/bash
# tolerance of the node relaxations (complementarity relative to the objective) against the bench value
run() { echo "== $*"; env "$@" python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s nodes %d iterations %d roofline %.3f group ms %.2f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), c['bnb_nodes'], c['ipm_iterations'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms', 0)))
"; }
run MIQP_QPTOL=3e-6 MIQP_QPTOL_F=3e-4
run MIQP_QPTOL=1e-5 MIQP_QPTOL_F=1e-3
run MIQP_QPTOL=3e-7 MIQP_QPTOL_F=3e-5

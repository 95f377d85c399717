#!/bin/bash
# the larger variant of the on-chip kernel (MIQP_OC_BIG=0: off) with and without slack front-point rows in the probes
for v in "MIQP_OC_BIG=0 MIQP_PROBE_MARGIN=0" "MIQP_OC_BIG=1 MIQP_PROBE_MARGIN=0" "MIQP_OC_BIG=1 MIQP_PROBE_MARGIN=0.5" "MIQP_OC_BIG=1 MIQP_PROBE_MARGIN=0.25"; do
  echo "== $v"
  env $v python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s nodes %d roofline %.3f launch ms %.2f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), c['bnb_nodes'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms', 0)))
"
  env $v python tools/single_latency.py 96 0.1 | tail -n 1
done

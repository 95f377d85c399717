#!/bin/bash
# rounding probes without the front-point environment / obstacle rows that hold with room to spare: bench value, launch times, single-solve latency
for m in 0 0.25 0.5 1.0 2.0; do
  echo "== MIQP_PROBE_MARGIN=$m"
  MIQP_PROBE_MARGIN=$m python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s nodes %d roofline %.3f launch ms %.2f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), c['bnb_nodes'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms', 0)))
"
  MIQP_PROBE_MARGIN=$m python tools/single_latency.py 96 0.1 | tail -n 1
done

#!/bin/bash
# rounding probes eligible every k-th round only: bench value, nodes and the launch times of the interior point kernels
for k in 1 2 3 4 6; do
  echo "== MIQP_PROBE_EVERY=$k"
  MIQP_PROBE_EVERY=$k python bench.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
c = d['config']
print('value %.1f ms/step %.0f proven %s/%s nodes %s roofline %.3f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), c.get('nodes_total', c.get('nodes')), d['roofline']['frac']))
print({k: v for k, v in d['roofline'].items() if k not in ('traffic',)})
"
done

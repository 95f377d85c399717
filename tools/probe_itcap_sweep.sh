#!/bin/bash
# unconverged rounding probes abandoned after k iterations: bench value and proven count
for k in 0 24 18 14; do
  echo "== MIQP_PROBE_ITCAP=$k"
  MIQP_PROBE_ITCAP=$k python bench.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
c = d['config']
print('value %.1f ms/step %.0f proven %s/%s roofline %.3f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), d['roofline']['frac']))
"
done

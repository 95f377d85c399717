#!/bin/bash
# lanes of the streaming call (contexts that share the device) against the bench value
for l in 1 2 3 4; do
  echo "== MIQP_LANES=$l"
  MIQP_LANES=$l python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s roofline %.3f launch ms %.2f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), d['roofline']['frac'], d['roofline'].get('avg_launch_ms', 0)))
"
done

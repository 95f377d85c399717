#!/bin/bash
# resident wavefronts of the concurrent launch of the larger on-chip variant against the bench value
for g in 1024 768 512 384 256; do
  echo "== MIQP_BIG_GRID=$g"
  MIQP_BIG_GRID=$g python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s nodes %d roofline %.3f launch ms %.2f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), c['bnb_nodes'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms', 0)))
"
done

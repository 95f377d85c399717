#!/bin/bash
# in-flight settings between 1024 and 2048: value and share proven (the default is the largest setting that proves >= 99 %)
for b in 1280 1536 1792; do
  python bench.py --batch $b --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('in flight $b: %.1f solves/s, proven %d / %d = %.4f, nodes per instance %d, latency p95 %.2f s' % (d['value'], c['instances_solved_to_gap'], c['instances_attempted'], c['instances_solved_to_gap']/c['instances_attempted'], c['bnb_nodes']/c['instances_attempted'], c['solve_latency_s_rank0']['p95']))"
done

#!/bin/bash
# resident wavefronts of the concurrent larger-variant launch: time per node of the same queue (two runs each)
for g in 1024 640 384; do
  for k in 1 2; do
    MIQP_BIG_GRID=$g python tools/stream_check.py 4096 1280 0 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d['timing']
print('grid $g: rounds loop %.2f s, nodes %d, %.1f ns per node, ipm %.2f s over %d rounds' % (t['solve_s'], d['nodes'], 1e9 * t['solve_s'] / d['nodes'], t['ipm_s'], t['ipm_launches']))"
  done
done

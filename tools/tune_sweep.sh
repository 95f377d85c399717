#!/bin/bash
# ./tune_sweep.sh STEP lib...: one 256-instance bench step per library variant (solve time, nodes, iterations per node)
st=$1; shift
for l in "$@"; do
  echo "== $l"; MIQP_GPU_LIB=$PWD/planner_miqp_amd/libmiqp_gpu$l.so BP_TOP=0 python tools/batch_profile.py $st 2>&1 | grep "^batch" | python3 -c "
import sys, json, re
for ln in sys.stdin:
    t = json.loads(ln[ln.index('{'):]); m = re.match(r'batch ([\d.]+) s, (\d+) nodes, solved (\d+)', ln)
    print('solve_s %.2f solved %s nodes %.2fM it/node %.2f ipm_s %.2f ns/node-it %.1f' % (t['solve_s'], m.group(3), t['nodes'] / 1e6, t['ipm_iters'] / t['nodes'], t['ipm_s'], 1e9 * t['ipm_s'] / t['ipm_iters']))
"
done

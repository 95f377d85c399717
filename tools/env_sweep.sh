#!/bin/bash
# ./env_sweep.sh STEP VAR value...: one 256-instance bench step per value of an environment switch
st=$1; var=$2; shift 2
for v in "$@"; do
  echo "== $var=$v"; env $var=$v BP_TOP=0 python tools/batch_profile.py $st 2>&1 | grep "^batch\|quantiles" | python3 -c "
import sys, json, re
for ln in sys.stdin:
    if ln.startswith('batch'):
        t = json.loads(ln[ln.index('{'):]); m = re.match(r'batch ([\d.]+) s, (\d+) nodes, solved (\d+)', ln)
        print('solve_s %.2f solved %s nodes %.2fM it/node %.2f ipm_s %.2f' % (t['solve_s'], m.group(3), t['nodes'] / 1e6, t['ipm_iters'] / t['nodes'], t['ipm_s']))
    else: print(ln.strip()[:140])
"
done

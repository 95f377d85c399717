#!/bin/bash
# warm start of the node relaxations (GPU): iterations per node, wall time, node count on a streaming queue
run() { echo "== $*"; env "$@" python tools/stream_check.py 2048 512 2>&1 | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['timing']
print('solved %d/%d in %.2f s, %.2fM nodes, %.2f iterations per node, ipm %.2f s, %.1f ns per node-iteration'%(d['solved'],d['Q'],t['solve_s'],d['nodes']/1e6,t['ipm_iters']/max(1,t['nodes']),t['ipm_s'],1e9*t['ipm_s']/max(1,t['ipm_iters'])))"; }
run MIQP_WARM=0
run MIQP_WARM=1
run MIQP_WARM=1 MIQP_WS_MU=0.1
run MIQP_WARM=1 MIQP_WS_MU=10
run MIQP_WARM=1 MIQP_WS_MU=1 MIQP_WS_DELTA=1e-1
run MIQP_WARM=1 MIQP_WS_MU=1 MIQP_WS_DELTA=1e-3
run MIQP_WARM=1 MIQP_WS_MU=0.01 MIQP_WS_DELTA=1e-3

#!/bin/bash
# probes only at nodes whose dual value leaves room below the incumbent (GPU): wall time, nodes, share of the memory-backed kernel
for r in 0 1 2 4 8; do
  echo "== MIQP_PROBE_ROOM=$r"
  MIQP_PROBE_ROOM=$r MIQP_STATS=1 python tools/stream_check.py 2048 256 2>&1 | grep "^{\|on-chip nodes" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d['timing']; print('solved %d/%d in %.2f s, %.2fM nodes, ipm %.2f s, launches %d, p95 %.2f max %.2f'%(d['solved'],d['Q'],t['solve_s'],d['nodes']/1e6,t['ipm_s'],t['ipm_launches'],d['latency']['95'],d['latency']['100']))
    else: print(l[:140])"
done

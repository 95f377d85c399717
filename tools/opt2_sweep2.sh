#!/bin/bash
# rounding-probe policy sweep (GPU): wall time and nodes of a streaming queue
for o in 128 64 32 0 $((128+65536)) $((64+65536)); do
  echo "== MIQP_OPT2=$o"
  MIQP_OPT2=$o python tools/stream_check.py 4096 1024 2>&1 | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['timing']
print('solved %d/%d in %.2f s, %.2fM nodes, ipm %.2f s, launches %d, p95 %.2f max %.2f'%(d['solved'],d['Q'],t['solve_s'],d['nodes']/1e6,t['ipm_s'],t['ipm_launches'],d['latency']['95'],d['latency']['100']))"
done

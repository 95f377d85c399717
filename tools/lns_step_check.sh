#!/bin/bash
cd "$(dirname "$0")/.."
for st in ${STEPS:-0 1e-4 5e-4 2e-3}; do
  echo "== MIQP_LNS_STEP=$st"
  MIQP_LNS_STEP=$st python tools/hard_trace.py cfg3 1059 1913 243 307 118 712 2>/dev/null | grep "==" | cut -c1-110
  MIQP_LNS_STEP=$st TL=10 python tools/hard_trace.py cfg5 2 9 14 15 11 5 2>/dev/null | grep "==" | cut -c1-110
  MIQP_LNS_STEP=$st python tools/stream_check.py 2048 256 0 10 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream 2048@256: solve_s %.3f nodes %d solved %d' % (d['timing']['solve_s'], d['nodes'], d['solved']))"
  MIQP_LNS_STEP=$st python tools/cfg5_check.py crowd | cut -c1-100
done

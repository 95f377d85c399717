#!/bin/bash
# local search modes: node counts of hard single instances and a short stream
cd "$(dirname "$0")/.."
for m in ${MODES:-13 45}; do
  echo "== MIQP_LNS=$m"
  MIQP_LNS=$m python tools/hard_trace.py cfg3 1059 1008 1913 243 307 118 1580 712 938 662 2>/dev/null | grep "=="
  MIQP_LNS=$m TL=10 python tools/hard_trace.py cfg5 2 9 14 15 11 5 2>/dev/null | grep "=="
  MIQP_LNS=$m python tools/stream_check.py 2048 256 0 10 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream 2048@256: solve_s %.3f nodes %d solved %d p99 %.2f max %.2f' % (d['timing']['solve_s'], d['nodes'], d['solved'], d['latency']['99'], d['latency']['100']))"
done

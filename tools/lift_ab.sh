#!/bin/bash
# relaxed front-point rows on / off (MIQP_RELAX_FRONT=0) (GPU): hard single seeds at rounds of 4096 nodes, a streaming queue, cfg4, cfg5
for rf in 1 0; do
  echo "== MIQP_RELAX_FRONT=$rf"
  MIQP_RELAX_FRONT=$rf WIDTHS=4096 python tools/width_probe.py 118 307 503 179 165 20 2>&1 | tail -n 6
  MIQP_RELAX_FRONT=$rf MIQP_STATS=1 python tools/stream_check.py 2048 512 2>&1 | grep "^{\|node outcomes:\|region branchings" | cut -c1-700
  MIQP_RELAX_FRONT=$rf python tools/stream_check.py 256 256 1000 10 cfg4 2>&1 | tail -n 1 | cut -c1-330
  MIQP_RELAX_FRONT=$rf python tools/stream_check.py 16 16 0 10 cfg5 2>&1 | tail -n 1 | cut -c1-330
done

#!/bin/bash
# region-set tightening on / off (bit 15 of MIQP_SEQ_KINDS switches it off) (GPU): hard single seeds at rounds of 4096 nodes and a streaming queue
for sk in 1280 $((1280 + 0x8000)); do
  echo "== MIQP_SEQ_KINDS=$sk"
  MIQP_SEQ_KINDS=$sk WIDTHS=4096 python tools/width_probe.py 118 307 503 179 165 20 2>&1 | tail -n 6
  MIQP_SEQ_KINDS=$sk MIQP_STATS=1 python tools/stream_check.py 1024 256 2>&1 | grep "^{\|node outcomes:\|region sets" | cut -c1-600
done

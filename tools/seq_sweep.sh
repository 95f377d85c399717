#!/bin/bash
# branching-mode sweep on hard seeds (GPU)
for sk in 1280 1288 1281 1289 $((1280+8+(4<<20))) $((1280+9+(4<<20))) $((1280 + (1<<17))); do
  echo "== MIQP_SEQ_KINDS=$sk"
  MIQP_SEQ_KINDS=$sk WIDTHS=4096 python tools/width_probe.py 118 307 503 179 165 2>&1 | tail -n 5
done

#!/bin/bash
# rounding-probe policy sweep (GPU), streaming queue 1024 / 256 in flight and hard single seeds
for o in 128 0 64 32 $((128+(4<<8))); do
  echo "== MIQP_OPT2=$o"
  MIQP_OPT2=$o python tools/stream_check.py 1024 256 2>&1 | tail -n 1 | cut -c1-400
  MIQP_OPT2=$o WIDTHS=4096 python tools/width_probe.py 118 307 179 2>&1 | tail -n 3
done

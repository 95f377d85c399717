#!/bin/bash
# run-to-run reproducibility of one small solve under the round-3 switches
for v in "MIQP_X=0" "MIQP_LIVE_INC=1" "MIQP_PROBE_OVERLAP=0" "MIQP_WARM=0" "MIQP_NOCUT=1"; do
  echo "== $v"
  env $v python tools/repeat_check.py 30 1e-3 2>&1 | tail -n 6
done

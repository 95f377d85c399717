#!/bin/bash
# warm start of the polish: complementarity / slack floor it is centred at, against the latency of single solves
for v in "MIQP_X=0" "MIQP_POLISH_MU=1e-3" "MIQP_POLISH_MU=1e-4" "MIQP_POLISH_MU=1e-4 MIQP_POLISH_DELTA=1e-5" "MIQP_POLISH_MU=1e-6 MIQP_POLISH_DELTA=1e-6" "MIQP_POLISH_COLD=1"; do
  echo "== $v"; env $v python tools/single_latency.py 96 0.1 | tail -n 1
done

#!/bin/bash
# two traced runs of the small test case, until a pair differs; prints the first differences
mkdir -p gpurun_out
for k in 1 2 3 4 5 6 7 8; do
  MIQP_TRACE=1 python tools/repeat_check.py 1 1e-3 2> gpurun_out/tr_$k.log | head -n 1
done
for k in 2 3 4 5 6 7 8; do
  if ! cmp -s gpurun_out/tr_1.log gpurun_out/tr_$k.log; then echo "== run 1 vs run $k"; diff gpurun_out/tr_1.log gpurun_out/tr_$k.log | head -n 30; break; fi
done

#!/bin/bash
# kernel-only A/B: the first full batch (32768 nodes) of the headline workload replayed under a timer, per library variant
for lib in "$@"; do
  echo "== $lib"; MIQP_GPU_LIB=$PWD/$lib MIQP_REPLAY=4 python bench.py --batch 256 --steps 1 --warmup 0 --no-cpu --time-limit 3 2>&1 | grep "replay"
done

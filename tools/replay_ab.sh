#!/bin/bash
# kernel A/B on the same batch: the first full batch of a bench stream replayed 6 times per library variant (MIQP_REPLAY)
cd "$(dirname "$0")/.."
for l in "$@"; do
  echo "== $l"
  MIQP_GPU_LIB=$PWD/tools/_build/$l MIQP_REPLAY=6 python bench.py --steps 1 --warmup 0 --no-cpu --no-extras --time-limit 3 2>&1 | grep -E "replay\]" | head -3
done

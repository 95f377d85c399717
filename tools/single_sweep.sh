#!/bin/bash
# single-solve latency (96 cfg3 seeds, gap 0.1) under a list of environment settings:  tools/single_sweep.sh "A=1" "B=2 C=3" ...
cd "$(dirname "$0")/.."
for e in "$@"; do echo "== $e"; env $e python tools/single_latency.py 96 0.1 2>/dev/null | cut -c1-420; done

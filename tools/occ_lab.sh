#!/bin/bash
# experiments of round 6 on a 10-step stream (tuning build; MIQP_* switches per run given as arguments, one run per argument)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/occ; mkdir -p $O
run() { env "$@" MIQP_GPU_LIB=tools/_build/libmiqp_gpu_tune.so python bench.py --steps 10 --warmup 0 --no-cpu --no-extras > $O/t.json 2> $O/t.err; echo "$*: $(python tools/bl.py $O/t.json)"; }
for w in "$@"; do run $w; done

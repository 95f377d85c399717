#!/bin/bash
# launch-split experiments of round 6 on a 10-step stream (tuning build; MIQP_* switches per run), then one round wavefront by wavefront (profile build)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/occ; mkdir -p $O
run() { env "$@" MIQP_GPU_LIB=tools/_build/libmiqp_gpu_tune.so python bench.py --steps 10 --warmup 0 --no-cpu --no-extras > $O/t.json 2> $O/t.err; echo "$*: $(python tools/bl.py $O/t.json)"; grep "larger block" $O/t.err | sed 's/.*in the larger block/    in the larger block/' | tail -1; }
run MIQP_STATS=1
run MIQP_BIG_W1=4 MIQP_BIG_W2=16
run MIQP_BIG_W1=3 MIQP_BIG_W2=24
MIQP_WAVE_DUMP=$O/waves.txt MIQP_GPU_LIB=tools/_build/libmiqp_gpu_prof.so python bench.py --steps 10 --warmup 0 --no-cpu --no-extras 2>&1 | grep "profile\]" | grep -i "active-set" | cut -c1-900; python tools/wave_dump.py $O/waves.txt | head -8

#!/bin/bash
# the measurements behind the launch split of a round: atomics on one word / adjacent words / words 4 KB apart, resident wavefronts per CU
# against the LDS block, and one round of a 10-step stream wavefront by wavefront - with the switches of the split off (full grids of the larger
# launches scanning the batch, unpadded LDS blocks, cold leaves of the local search) and as shipped  -> gpurun_out/split/
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/split; mkdir -p $O
tools/_build/atomic_lab 98304 > $O/atomic_lab.txt 2>&1
tools/_build/resident_lab > $O/resident_lab.txt 2>&1
P=tools/_build/libmiqp_gpu_prof.so
MIQP_CLS_LISTS=0 MIQP_BIG_PAD=0 MIQP_LNS_WARM=0 MIQP_WAVE_DUMP=$O/w_off.txt MIQP_GPU_LIB=$P python bench.py --steps 10 --warmup 0 --no-cpu --no-extras 2>&1 | grep "profile\]" | grep -i "standard active-set" | cut -c1-600 > $O/wave_dump_switches_off.txt
python tools/wave_dump.py $O/w_off.txt >> $O/wave_dump_switches_off.txt
MIQP_WAVE_DUMP=$O/w_on.txt MIQP_GPU_LIB=$P python bench.py --steps 10 --warmup 0 --no-cpu --no-extras 2>&1 | grep "profile\]" | grep -i "standard active-set" | cut -c1-600 > $O/wave_dump_shipped.txt
python tools/wave_dump.py $O/w_on.txt >> $O/wave_dump_shipped.txt
T=tools/_build/libmiqp_gpu_tune.so
run() { env "$@" MIQP_GPU_LIB=$T python bench.py --steps 10 --warmup 0 --no-cpu --no-extras > $O/t.json 2> $O/t.err; echo "$*: $(python tools/bl.py $O/t.json)"; grep "larger block" $O/t.err | sed 's/.*in the larger block/    in the larger block/' | tail -1; }
( run MIQP_STATS=1 MIQP_CLS_LISTS=0 MIQP_BIG_PAD=0 MIQP_LNS_WARM=0; run MIQP_STATS=1 MIQP_BIG_W1=0 MIQP_BIG_PAD=0 MIQP_LNS_WARM=0; run MIQP_STATS=1 MIQP_BIG_PAD=0 MIQP_LNS_WARM=0; run MIQP_STATS=1 MIQP_LNS_WARM=0; run MIQP_STATS=1; GPU_MAX_HW_QUEUES=4 run MIQP_STATS=1 ) > $O/ab_10_step_stream.txt 2>&1
rm -f $O/w_off.txt $O/w_on.txt $O/t.json $O/t.err
cat $O/ab_10_step_stream.txt; head -3 $O/wave_dump_switches_off.txt | cut -c1-300; head -3 $O/wave_dump_shipped.txt | cut -c1-300

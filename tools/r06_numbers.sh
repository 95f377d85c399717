#!/bin/bash
# the numbers of round 6 on one MI355X -> gpurun_out/r06/  (copied to profiles/r06_* afterwards).  tools/r06_numbers.sh [a|b|c]: the three parts
# fit one gpurun call each
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r06; mkdir -p $O
part=${1:-a}
if [ "$part" = a ]; then   # the bench lines
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_like_bench.json 2> $O/driver_like_bench.err
  python bench.py > $O/final_bench.json 2> $O/final_bench.err
  python bench.py --no-cpu --no-extras --steps 40 --warmup 0 > $O/long_stream_bench.json 2>/dev/null
  MIQP_AS=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $O/driver_like_bench_interior_point_only.json 2>/dev/null
  for f in driver_like_bench final_bench long_stream_bench driver_like_bench_interior_point_only; do python tools/bl.py $O/$f.json; done
fi
if [ "$part" = b ]; then   # the other configurations, single solves, heuristics
  python bench.py --config cfg4 --total 256 --batch 256 --steps 1 --warmup 1 --no-cpu --no-extras > $O/bench_cfg4_strong.json 2>/dev/null
  python tools/stream_check.py 256 256 0 10 cfg2 > $O/cfg2_256.json 2>/dev/null
  python tools/stream_check.py 256 256 1000 10 cfg4 > $O/cfg4_256.json 2>/dev/null
  python tools/stream_check.py 2048 256 0 10 cfg3 > $O/cfg3_2048_at_256.json 2>/dev/null
  python tools/cfg5_check.py both > $O/cfg5_check.txt 2>/dev/null
  (python tools/single_latency.py 96 0.1; python tools/single_latency.py 96 0.01) > $O/single_latency.txt 2>&1
  python tools/lns_ab.py 1913 662 118 712 243 307 1059 > $O/local_search_ab.txt 2>&1
  (for s in 3314; do python tools/lns_ab.py child $s; done) > $O/hard_single.txt 2>&1
  cat $O/cfg5_check.txt; cut -c1-330 $O/cfg2_256.json $O/cfg4_256.json $O/cfg3_2048_at_256.json; cut -c1-300 $O/bench_cfg4_strong.json; cat $O/single_latency.txt
fi
if [ "$part" = c ]; then   # profiles: kernel trace, counters, phase cycles
  tools/final_profiles.sh > $O/final_profiles.log 2>&1
  cp -r gpurun_out/final $O/final
  MIQP_GPU_LIB=tools/_build/libmiqp_gpu_prof.so python bench.py --steps 10 --warmup 0 --no-cpu --no-extras 2>&1 | grep "profile\]" > $O/phase_cycles_long_stream.txt
  MIQP_GPU_LIB=tools/_build/libmiqp_gpu_prof.so python bench.py --steps 1 --warmup 0 --no-cpu --no-extras --time-limit 3 2>&1 | grep "profile\]" > $O/phase_cycles_short.txt
  MIQP_STATS=1 python bench.py --steps 4 --warmup 1 --no-cpu --no-extras 2>&1 | grep "active-set" > $O/active_set_stats.txt
  tail -30 $O/final_profiles.log; cat $O/phase_cycles_long_stream.txt | cut -c 1-600
fi

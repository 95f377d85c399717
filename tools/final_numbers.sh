#!/bin/bash
# measurement campaign of the round (GPU): bench sweep over the instances in flight, other BASELINE configurations, single solves
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/final_numbers; mkdir -p $O
STEPS=4 tools/bench_sweep.sh > $O/batch_sweep.jsonl 2>$O/batch_sweep.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-stream --batch 1024 --steps 2 --warmup 1 --no-cpu > $O/bench_nostream_1024.json 2>/dev/null
python bench.py --no-stream --batch 256 --steps 2 --warmup 1 --no-cpu > $O/bench_nostream_256.json 2>/dev/null
python bench.py --config cfg4 --total 256 --batch 256 --steps 1 --warmup 1 --no-cpu > $O/bench_cfg4_strong.json 2>/dev/null
python tools/stream_check.py 256 256 0 10 cfg2 > $O/cfg2.json 2>/dev/null
python tools/stream_check.py 256 256 1000 10 cfg4 > $O/cfg4.json 2>/dev/null
python tools/stream_check.py 16 16 0 10 cfg5 > $O/cfg5.json 2>/dev/null
python tools/single_latency.py 96 0.01 > $O/single_latency.txt 2>/dev/null
python tools/single_latency.py 96 0.1 >> $O/single_latency.txt 2>/dev/null
python tools/single_detail.py 96 0.1 | tail -n 1 >> $O/single_latency.txt 2>/dev/null
cat $O/batch_sweep.jsonl | cut -c1-260; cut -c1-600 $O/bench_default.json; cat $O/single_latency.txt; cut -c1-300 $O/cfg4.json $O/cfg5.json $O/cfg2.json; cut -c1-500 $O/bench_cfg4_strong.json

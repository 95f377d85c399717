#!/bin/bash
# kernel trace of two bench steps -> gpurun_out/trace_now/{kernel_stats.csv,round_gaps.txt}  (tools/trace_now.sh [tag])
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/trace_${1:-now}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu --no-extras > $O/trace_bench.log 2>&1
python3 $R/tools/round_gaps.py $O/trace > $O/round_gaps.txt 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep "^{" $O/trace_bench.log > $O/trace_bench.json
rm -rf $O/trace
head -12 $O/kernel_stats.csv; cat $O/round_gaps.txt

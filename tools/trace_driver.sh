#!/bin/bash
# rocprofv3 kernel trace + stats of the driver's command (without the CPU leg and the extra legs) -> gpurun_out/trace_driver/
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/trace_driver; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $O/trace_bench.log 2>&1
python3 $R/tools/round_gaps.py $O/trace timeline > $O/round_gaps.txt 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep "^{" $O/trace_bench.log > $O/trace_bench.json
rm -rf $O/trace
head -9 $O/kernel_stats.csv; cat $O/round_gaps.txt; python3 $R/tools/bl.py $O/trace_bench.json

#!/bin/bash
# kernel stats of ONE single solve with a time limit: tools/trace_one.sh cfg seed limit [tag]
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/trace_one_${4:-x}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
VERBOSE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o one -- python3 $R/tools/one.py $1 $2 $3 > $O/one.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
head -3 $O/one.log | cut -c 1-200; head -10 $O/kernel_stats.csv

#!/bin/bash
# profile set of the 4-car configuration (cfg5: 4 cars x 30 steps x 64 regions, ipm_kernel<4,128>): kernel trace + stats and one SQ counter pass
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/cfg5prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o c5 -- python3 $R/tools/stream_check.py 16 16 0 10 cfg5 > $O/run.json 2>$O/run.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_SQ1 -o p -- python3 $R/tools/stream_check.py 16 16 0 3 cfg5 > $O/pmc1.json 2>$O/pmc1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_SQ2 -o p -- python3 $R/tools/stream_check.py 16 16 0 3 cfg5 > $O/pmc2.json 2>$O/pmc2.err
cd $R
python3 tools/pmc_traffic.py $O/pmc_SQ1 $O/pmc_SQ2 > $O/sq_raw.json 2>$O/sq_raw.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
head -6 $O/kernel_stats.csv; cut -c1-300 $O/run.json

#!/bin/bash
# Collects the round's profile artefacts on the GPU box into gpurun_out/final/ (copied to profiles/ afterwards):
#   kernel trace + stats of one default bench step (1280 in flight, queue of 2560), three separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters;
#   no tracing together with --pmc) on `bench.py --steps 1 --warmup 0 --no-cpu --time-limit 3` for the counters.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu --no-extras > $O/trace_bench.log 2>&1
python3 $R/tools/round_gaps.py $O/trace > $O/round_gaps.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --time-limit 3 > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_SQ1 -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --time-limit 3 > $O/pmc_SQ1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_SQ2 -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --time-limit 3 > $O/pmc_SQ2.log 2>&1
# (a third SQ pass, may be refused on a box whose rocprofv3 does not know a counter: its absence is not an error)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_SQ3 -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-extras --time-limit 3 > $O/pmc_SQ3.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/pmc_SQ3 > $O/sq3_raw.json 2>$O/sq3_raw.err
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/traffic_raw.json 2>$O/traffic_raw.err
python3 tools/pmc_traffic.py $O/pmc_SQ1 $O/pmc_SQ2 > $O/sq_raw.json 2>$O/sq_raw.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep "^{" $O/trace_bench.log > $O/trace_bench.json
rm -rf $O/trace/*/*_kernel_trace.csv $O/pmc_*/*/*counter_collection.csv 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -delete 2>/dev/null; find $O -name "*counter_collection.csv" -delete 2>/dev/null
ls -la $O | head -30; head -8 $O/kernel_stats.csv

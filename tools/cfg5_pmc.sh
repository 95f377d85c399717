#!/bin/bash
# SQ counters of ipm_kernel<4, NT> over one cfg5 solve (seed 15, ~3 s): tools/gpu.sh 900 'bash tools/cfg5_pmc.sh'  -> gpurun_out/cfg5_pmc/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/cfg5_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -o p -- python3 $R/tools/cfg5_rounds.py 15 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -o p -- python3 $R/tools/cfg5_rounds.py 15 > $O/b.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $O/c -o p -- python3 $R/tools/cfg5_rounds.py 15 > $O/c.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC --output-format csv -d $O/d -o p -- python3 $R/tools/cfg5_rounds.py 15 > $O/d.log 2>&1
cd $R; python3 - <<'PY' > $O/summary.txt
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/cfg5_pmc/*/*/*counter_collection.csv") + glob.glob("gpurun_out/cfg5_pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ipm_kernel<4" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-28s %16.0f  (%d launches)" % (k, tot[k], n[k]))
PY
cat $O/summary.txt; tail -2 $O/a.log
rm -rf $O/*/*/*counter_collection.csv $O/*/*counter_collection.csv
# HBM traffic of the kernel (separate passes, as the guide prescribes): bytes = FETCH_SIZE x 64 B x 2 (gfx950: FETCH_SIZE reports half of a wide streaming read), WRITE_SIZE x 64 B
if [ "$1" = "traffic" ]; then
  cd /tmp; O=$R/gpurun_out/cfg5_pmc
  for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $O/t_$c -o p -- python3 $R/tools/cfg5_rounds.py 15 > $O/t_$c.log 2>&1; done
  cd $R; python3 - <<'PY' >> $O/summary.txt
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("gpurun_out/cfg5_pmc/t_*/*/*counter_collection.csv") + glob.glob("gpurun_out/cfg5_pmc/t_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ipm_kernel<4" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-28s %16.0f  (%d launches)" % (k, tot[k], n[k]))
PY
  tail -3 $O/summary.txt; rm -rf $O/t_*/*/*counter_collection.csv $O/t_*/*counter_collection.csv
fi

#!/bin/bash
# bench.py (default stream) under a list of environment settings, one summary line each:  tools/bench_matrix.sh outdir "A=1 B=2" "A=2" ...
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; shift; mkdir -p $OUT
k=0
for e in "$@"; do
  k=$((k+1))
  env $e python bench.py --no-cpu --no-extras ${BENCH_ARGS} > $OUT/m$k.json 2> $OUT/m$k.err
  python - "$e" $OUT/m$k.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
    print("%-44s value %7.1f  proven %.4f  nodes/inst %6.0f  it/node %5.2f  launch %6.2f ms x %4d  frac %.4f  p95 %.2f s" % (sys.argv[1], d["value"], c["instances_solved_to_gap"] / c["instances_attempted"],
          c["bnb_nodes"] / c["instances_attempted"], c["ipm_iterations"] / max(1, c["bnb_nodes"]), r["avg_launch_ms"], r["launches"], r["frac"], c["solve_latency_s_rank0"]["p95"]))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
done

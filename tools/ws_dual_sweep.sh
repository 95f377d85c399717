#!/bin/bash
# sweep of the warm start from the parent's multipliers (MIQP_WS_DUAL / THETA / MU / DELTA) on a short bench stream; prints one summary line per setting
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-wsd}; mkdir -p $OUT
run() {  # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 2 --warmup 1 --no-cpu --no-extras --batch ${BATCH:-1024} > $OUT/$name.json 2> $OUT/$name.err
  python - "$name" $OUT/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
    print("%-28s value %7.1f  proven %.4f  nodes/inst %7.0f  it/node %5.2f  launch %6.2f ms x %d  frac %.4f" % (sys.argv[1], d["value"], c["instances_solved_to_gap"] / c["instances_attempted"],
          c["bnb_nodes"] / c["instances_attempted"], c["ipm_iterations"] / max(1, c["bnb_nodes"]), r["avg_launch_ms"], r["launches"], r["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run dual0 MIQP_WS_DUAL=0
run dual1_t1 MIQP_WS_DUAL=1
run dual1_t05 MIQP_WS_DUAL=1 MIQP_WS_THETA=0.5
run dual1_t02 MIQP_WS_DUAL=1 MIQP_WS_THETA=0.2
run dual1_t0 MIQP_WS_DUAL=1 MIQP_WS_THETA=0.0
run dual1_t1_mu01 MIQP_WS_DUAL=1 MIQP_WS_MU=0.1
run dual1_t02_mu01 MIQP_WS_DUAL=1 MIQP_WS_THETA=0.2 MIQP_WS_MU=0.1
run dual1_t02_mu001 MIQP_WS_DUAL=1 MIQP_WS_THETA=0.2 MIQP_WS_MU=0.01
run dual1_t02_mu01_d2 MIQP_WS_DUAL=1 MIQP_WS_THETA=0.2 MIQP_WS_MU=0.1 MIQP_WS_DELTA=0.01
run dual1_t05_mu01_d2 MIQP_WS_DUAL=1 MIQP_WS_THETA=0.5 MIQP_WS_MU=0.1 MIQP_WS_DELTA=0.01
run dual0_t02 MIQP_WS_DUAL=0 MIQP_WS_THETA=0.2

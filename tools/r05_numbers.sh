#!/bin/bash
# the other BASELINE configurations at the end of round 5 (the bench lines themselves: profiles/r05_driver_like_bench_final_commit.json)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r05_numbers; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --config cfg4 --total 256 --batch 256 --steps 1 --warmup 1 --no-cpu --no-extras > $O/bench_cfg4_strong.json 2>/dev/null
python tools/stream_check.py 256 256 0 10 cfg2 > $O/cfg2.json 2>/dev/null
python tools/stream_check.py 256 256 1000 10 cfg4 > $O/cfg4.json 2>/dev/null
python tools/stream_check.py 2048 256 0 10 cfg3 > $O/cfg3_2048_at_256.json 2>/dev/null
python tools/cfg5_check.py both > $O/cfg5.txt 2>/dev/null
python bench.py --no-cpu --no-extras --steps 40 --warmup 0 > $O/bench_long_stream.json 2>/dev/null
cat $O/cfg5.txt; cut -c1-330 $O/cfg2.json $O/cfg4.json $O/cfg3_2048_at_256.json; cut -c1-400 $O/bench_cfg4_strong.json
python - <<'PY'
import json
for f in ("bench_default", "bench_long_stream"):
    d = json.loads(open("gpurun_out/r05_numbers/%s.json" % f).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
    print(f, "value %.1f proven %.4f frac %.4f launch %.2f ms x %d nodes/inst %.0f it/node %.2f attempted %d" % (d["value"], d["proven_share"], r["frac"], r["avg_launch_ms"], r["launches"], c["bnb_nodes"] / c["instances_attempted"], c["ipm_iterations"] / c["bnb_nodes"], c["instances_attempted"]))
PY

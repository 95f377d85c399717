#!/bin/bash
# second measurement campaign of round 4 (after the re-rounding of infeasible probes): as tools/r04_numbers.sh without the long extra legs
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/r04b_numbers; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $O/bench_driver_like.json 2> $O/bench_driver_like.err
python bench.py --config cfg4 --total 256 --batch 256 --steps 1 --warmup 1 --no-cpu > $O/bench_cfg4_strong.json 2>/dev/null
python tools/stream_check.py 256 256 0 10 cfg2 > $O/cfg2.json 2>/dev/null
python tools/stream_check.py 256 256 1000 10 cfg4 > $O/cfg4.json 2>/dev/null
python tools/stream_check.py 2048 256 0 10 cfg3 > $O/cfg3_2048_at_256.json 2>/dev/null
python tools/cfg5_check.py both > $O/cfg5.txt 2>/dev/null
python tools/single_latency.py 96 0.01 > $O/single_latency.txt 2>/dev/null
python tools/single_latency.py 96 0.1 >> $O/single_latency.txt 2>/dev/null
cut -c1-900 $O/bench_default.json; echo; python - <<'PY'
import json
for f in ("bench_default", "bench_driver_like"):
    d = json.loads(open("gpurun_out/r04b_numbers/%s.json" % f).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
    print(f, "value %.1f proven %.4f frac %.4f launch %.2f ms x %d nodes/inst %.0f it/node %.2f" % (d["value"], d["proven_share"], r["frac"], r["avg_launch_ms"], r["launches"], c["bnb_nodes"] / c["instances_attempted"], c["ipm_iterations"] / c["bnb_nodes"]))
    for k in ("value_all_proven", "all_proven", "in_flight_sweep", "one_batch_control", "cpu_baseline"):
        if k in d: print("   ", k, d[k])
PY
cat $O/cfg5.txt $O/single_latency.txt; cut -c1-330 $O/cfg2.json $O/cfg4.json $O/cfg3_2048_at_256.json; cut -c1-400 $O/bench_cfg4_strong.json

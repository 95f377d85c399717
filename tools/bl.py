"""one line of a bench.py JSON file: python tools/bl.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r = d["roofline"]
ts = d["config"].get("timed_stream_rank0") or {}
print("value %.1f ms_per_step %.1f proven %.5f launch_ms %.3f launches %d nodes %s iters %s frac %.4f; last admission %.2f s, %.0f solves/s until then" % (d["value"], d["ms_per_step"], d.get("proven_share", 0), r["avg_launch_ms"], r["launches"], d["config"].get("bnb_nodes"), d["config"].get("ipm_iterations"), r["frac"], ts.get("last_admission_s", 0), ts.get("solves_per_s_with_backlog", 0)))

"""one line of a bench.py JSON file: python tools/bl.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r = d["roofline"]
print("value %.1f ms_per_step %.1f proven %.5f launch_ms %.3f launches %d nodes %s iters %s frac %.4f" % (d["value"], d["ms_per_step"], d.get("proven_share", 0), r["avg_launch_ms"], r["launches"], d["config"].get("bnb_nodes"), d["config"].get("ipm_iterations"), r["frac"]))

import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r=d["roofline"]; ts=d["config"].get("timed_stream_rank0") or {}
print("value %.1f proven %.5f std %.3f ms group %.3f ms x %d; last admission %.2f s, %.0f/s" % (d["value"], d.get("proven_share",0), r["avg_launch_ms"], r["launch_group_ms"], r["launches"], ts.get("last_admission_s",0), ts.get("solves_per_s_with_backlog",0)))

"""Latency of single solves (one callCplex per instance, the reference's usage pattern) on cfg3 seeds (GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
lat = []; solved = 0; per = []
w = P.CplexWrapper()
for s in range(n):
    p = synthetic.generate("cfg3", s, gap=gap, max_time=10.0)
    w.resetParameters(p)
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    pr = w.getSolutionProperties()
    ok = int(st) == 0 and pr.status in (101, 102)
    solved += ok
    if ok and s > 0:
        lat.append(dt); per.append((dt, s, int(pr.nodes), int(w.lastTiming()['ipm_launches'])))
print("gap", gap, "solved", solved, "of", n, "latency ms p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(1e3 * np.percentile(lat, q) for q in (50, 90, 99, 100)))
print("slowest:", ["seed %d: %.1f ms, %d nodes, %d rounds" % (b, 1e3 * a, c, d) for a, b, c, d in sorted(per, reverse=True)[:6]])

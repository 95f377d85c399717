"""per-instance view of single solves (GPU): wall time, rounds, nodes, iterations, time inside the interior point launches"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
w = P.CplexWrapper()
rows = []
for s in range(n):
    p = synthetic.generate("cfg3", s, gap=gap, max_time=10.0)
    w.resetParameters(p)
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    pr = w.getSolutionProperties(); tm = w.lastTiming()
    rows.append((dt * 1e3, tm["ipm_launches"], pr.nodes, tm["ipm_iters"], tm["ipm_s"] * 1e3, tm["solve_s"] * 1e3, pr.time * 1e3))
rows = rows[1:]
rows.sort()
for r in rows[::max(1, len(rows) // 16)]:
    print("wall %.2f ms: rounds %d nodes %d iterations %d, ipm launches %.2f ms, device span %.2f ms, rounds loop %.2f ms" % r)
a = np.array(rows)
print("median wall %.2f ms, rounds %.0f, ipm %.2f ms, device span %.2f ms, per round %.3f ms" % (np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 4]), np.median(a[:, 5]), np.median(a[:, 5] / np.maximum(1, a[:, 1]))))

"""4-car instances through the wide (3-4 car) interior point path, with phase shares from the profile build (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
for cfg, n, tl in (("cfg5s", 16, 3.0), ("cfg5", 1, 4.0)):
    ws = []
    for s in range(n):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate(cfg, s, gap=0.01, max_time=tl)); ws.append(w)
    t = time.time(); sts = P.solve_batch(ws); dt = time.time() - t
    tm = ws[0].lastTiming()
    print(cfg, "instances", n, "wall %.2f s" % dt, "nodes", tm["nodes"], "iters", tm["ipm_iters"], "ipm_s %.2f" % tm["ipm_s"], "nodes/s %.0f" % (tm["nodes"] / max(tm["ipm_s"], 1e-9)),
          "solved", sum(int(s) == 0 and w.getSolutionProperties().status in (101, 102) for s, w in zip(sts, ws)))

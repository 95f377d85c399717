"""how the incumbent a cfg5 instance holds at the limit differs from its optimum (GPU only): the car/car side sequences (rear/rear group), the region
sequences and the distance of the cars from each other, for two time limits.  python tools/cfg5_incumbents.py seed [short_limit long_limit]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
seed = int(sys.argv[1]); tls = [float(a) for a in sys.argv[2:4]] or [10.0, 45.0]
res = []
for tl in tls:
    w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg5", seed, gap=0.01, max_time=tl))
    t = time.time(); st = w.callCplex(); dt = time.time() - t; pr = w.getSolutionProperties(); r = w.getRawResults()
    print("limit %.0f s: status %d obj %.3f bound %.3f nodes %d time %.2f" % (tl, pr.status, pr.objective, pr.best_bound, pr.nodes, dt), flush=True)
    res.append(r)
Cn, N, R, E, O, L = res[0].dims; K = Cn - 1
def side_seq(r, c1, c2, g):
    a = r.car2car_collision[c1, c2 - 1, :, 4 * g:4 * g + 4]
    return "".join("." if a[i].sum() == 0 else str(int(np.argmin(a[i]))) for i in range(N))
for c1 in range(Cn):
    for c2 in range(c1 + 1, Cn):
        for g in range(4):
            s = [side_seq(r, c1, c2, g) for r in res]
            print("pair %d-%d group %d: %s" % (c1, c2, g, "  |  ".join(s)) + ("   DIFFERENT" if len(set(s)) > 1 else ""))
for c in range(Cn):
    for r in res:
        reg = np.argmax(r.active_region[c], axis=1)
        print("car %d regions %s  x %s  y %s  vx %s" % (c, " ".join("%d" % v for v in reg), np.round(r.pos_x[c, ::5], 1), np.round(r.pos_y[c, ::5], 1), np.round(r.vel_x[c, ::5], 1)))

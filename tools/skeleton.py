"""the discrete skeleton of solutions of an instance found early and late: regions per car and step, relative position of the cars.
python tools/skeleton.py cfg seed"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg, seed = sys.argv[1], int(sys.argv[2])
def show(tag, w):
    pr = w.getSolutionProperties(); r = w.getRawResults()
    print("== %s: status %d obj %.3f bound %.3f nodes %d" % (tag, pr.status, pr.objective, pr.best_bound, pr.nodes))
    C, N = r.pos_x.shape
    reg = r.active_region.argmax(-1)
    for c in range(C):
        print("  car %d region : %s" % (c, " ".join("%2d" % v for v in reg[c])))
        print("        x      : %s" % " ".join("%5.1f" % v for v in r.pos_x[c]))
        print("        y      : %s" % " ".join("%5.2f" % v for v in r.pos_y[c]))
        print("        vx     : %s" % " ".join("%5.2f" % v for v in r.vel_x[c]))
    for a in range(C):
        for b in range(a + 1, C):
            print("  pair %d-%d dx : %s" % (a, b, " ".join("%5.1f" % v for v in (r.pos_x[b] - r.pos_x[a]))))
            print("           dy : %s" % " ".join("%5.2f" % v for v in (r.pos_y[b] - r.pos_y[a])))
for tl in [float(x) for x in os.environ.get("TLS", "0.03,0.1,0.3,40").split(",")]:
    p = synthetic.generate(cfg, seed, gap=0.01, max_time=tl)
    w = P.CplexWrapper(); w.resetParameters(p)
    st = w.callCplex()
    if int(st) == 0:
        show("limit %.2f s" % tl, w)
    else:
        print("== limit %.2f s: status %d" % (tl, int(st)))

"""Single solves of synthetic instances, one line each (A/B of the active-set launch: run once with MIQP_AS=0 and once with MIQP_AS=1
and compare the objectives).  python tools/as_check.py [cfg] [nseeds] [gap] [first]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
gap = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tot = 0.0
for s in range(first, first + n):
    w = P.CplexWrapper(); w.resetParameters(synthetic.generate(cfg, s, gap=gap, max_time=20)); t = time.time(); st = w.callCplex(); dt = time.time() - t
    pr = w.getSolutionProperties(); tot += dt
    print("seed %4d status %d/%d obj %.9f bound %.9f nodes %7d iters %9d time %.4f" % (s, int(st), pr.status, pr.objective, pr.best_bound, pr.nodes, pr.NrIterations, dt), flush=True)
print("total %.3f s" % tot)

"""progress of single hard instances (bound / incumbent over time): python tools/hard_trace.py cfg seed [seed ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg = sys.argv[1]
for s in sys.argv[2:]:
    w = P.CplexWrapper(verbose=1); w.resetParameters(synthetic.generate(cfg, int(s), gap=0.01, max_time=float(os.environ.get("TL", "20"))))
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    pr = w.getSolutionProperties()
    print("== %s seed %s: status %d gap %.4f obj %.3f bound %.3f nodes %d time %.2f pool %d" % (cfg, s, pr.status, pr.gap, pr.objective, pr.best_bound, pr.nodes, dt, pr.NrSolutionPool), flush=True)

"""progress trace of single hard instances: python tools/trace_one.py cfg3 307 30 [712 ...] (seed list, last number = time limit)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg = sys.argv[1]; seeds = [int(x) for x in sys.argv[2:-1]]; tl = float(sys.argv[-1])
for seed in seeds:
    w = P.CplexWrapper(verbose=1); w.resetParameters(synthetic.generate(cfg, seed, gap=0.01, max_time=tl))
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    pr = w.getSolutionProperties()
    print("seed %d status %d/%d gap %.5f obj %.4f bound %.4f nodes %d %.2f s" % (seed, int(st), pr.status, pr.gap, pr.objective, pr.best_bound, pr.nodes, dt), flush=True)

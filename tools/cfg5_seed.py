"""one cfg5 instance alone with a given limit: status, gap, nodes, time.  python tools/cfg5_seed.py seed limit"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11; lim = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg5", seed, gap=1e-2, max_time=lim)); st = int(w.callCplex()); pr = w.getSolutionProperties(); tm = w.lastTiming()
print("seed %d limit %.0f: status %d/%d objective %.6f bound %.6f gap %.4f nodes %d time %.2f s; rounds %d, kernel time %.2f s" % (seed, lim, st, pr.status, pr.objective, pr.best_bound, pr.gap, pr.nodes, pr.time, tm.get("ipm_launches", 0), tm.get("ipm_s", 0)))

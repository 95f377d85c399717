import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
for seed in (1913, 662, 307):
    w = P.CplexWrapper(); w.resetParameters(synthetic.generate('cfg3', seed, gap=1e-2, max_time=30)); st = int(w.callCplex()); pr = w.getSolutionProperties()
    print('RESULT', seed, st, pr.status, repr(pr.objective), pr.nodes, round(pr.time, 3), flush=True)

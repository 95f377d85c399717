import os, sys, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cap = int(sys.argv[1]); seeds = [int(x) for x in sys.argv[2:]] or [118]
for seed in seeds:
    w = P.CplexWrapper(max_open_nodes=cap, verbose=int(os.environ.get("VB", "0"))); w.resetParameters(synthetic.generate("cfg3", seed, gap=0.01, max_time=float(os.environ.get("TL", "6"))))
    t = time.time(); st = w.callCplex(); pr = w.getSolutionProperties()
    print(dict(seed=seed, st=int(st), status=pr.status, objective=pr.objective, bound=pr.best_bound, nodes=int(pr.nodes), t=round(time.time() - t, 2)), flush=True)

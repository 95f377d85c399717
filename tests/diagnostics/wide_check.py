"""3/4-car instances: device vs CPU oracle (diagnostic; GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
import planner_miqp_amd as P, oracle_lib
from planner_miqp_amd import synthetic
O = oracle_lib.Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
for cfg, seeds, gap in (("mini3", range(4), 1e-6), ("mini4", range(3), 1e-6), ("cfg5s", range(3), 1e-3)):
    for s in seeds:
        p = synthetic.generate(cfg, s, gap=gap, max_time=30)
        t = time.time(); w = P.CplexWrapper(); w.resetParameters(p); st = w.callCplex(); pr = w.getSolutionProperties(); tg = time.time() - t
        t = time.time(); h = O.from_params(p, 10); ost, r, op = O.solve(h, O.dims(p), gap=gap, time_limit=60); to = time.time() - t; O.free(h)
        print(cfg, s, "gpu st %d obj %.6f gap %.2e nodes %d (%.2fs) | oracle st %d obj %.6f gap %.2e (%.2fs)" % (int(st), pr.objective, pr.gap, pr.nodes, tg, ost, op.objective, op.gap, to), flush=True)

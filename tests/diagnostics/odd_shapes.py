"""Odd instance shapes (tiny / long horizons, no environment, 16 regions with 2 cars): device vs oracle (GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
import planner_miqp_amd as P, oracle_lib
from planner_miqp_amd import synthetic
O = oracle_lib.Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
bad = 0
for cfg in ((1, 2, 16, 1, 0), (1, 3, 32, 1, 0), (2, 2, 32, 1, 0), (2, 3, 16, 1, 0), (1, 40, 32, 1, 0), (1, 12, 32, 0, 0), (2, 6, 32, 0, 0), (2, 6, 16, 2, 1), (3, 4, 16, 1, 1)):
    for s in range(3):
        p = synthetic.generate(cfg, s, gap=1e-6, max_time=30)
        w = P.CplexWrapper(); w.resetParameters(p); st = w.callCplex(); pr = w.getSolutionProperties()
        h = O.from_params(p, 10); ost, r, op = O.solve(h, O.dims(p), gap=1e-6, time_limit=30); O.free(h)
        ok = int(st) == ost and (ost != 0 or op.gap > 2e-6 or abs(pr.objective - op.objective) <= 1e-5 * max(1.0, abs(op.objective)))
        bad += not ok
        print(cfg, s, "gpu", int(st), pr.objective, "oracle", ost, op.objective, "" if ok else "<<<< MISMATCH", flush=True)
print("mismatches", bad)

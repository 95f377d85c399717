"""One-off randomized parity sweep: device vs CPU oracle at a tight gap on many small multi-car instances (GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
import planner_miqp_amd as P, oracle_lib
from planner_miqp_amd import synthetic
O = oracle_lib.Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
bad = 0; n = 0; t0 = time.time()
OFF = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for cfg, seeds in (("mini", range(6 + OFF, 70 + OFF)), ("mini3b", range(4 + OFF, 24 + OFF)), ("mini4b", range(4 + OFF, 14 + OFF)), ("mini1", range(6 + OFF, 30 + OFF)), ("cfg2", range(4 + OFF, 24 + OFF))):
    for s in seeds:
        if time.time() - t0 > float(sys.argv[1]) if len(sys.argv) > 1 else 400:
            break
        p = synthetic.generate(cfg, s, gap=1e-6, max_time=60)
        w = P.CplexWrapper(); w.resetParameters(p); st = w.callCplex(); pr = w.getSolutionProperties()
        h = O.from_params(p, 10); ost, r, op = O.solve(h, O.dims(p), gap=1e-6, time_limit=30); O.free(h)
        n += 1
        if op.gap > 2e-6:
            continue   # oracle did not finish: no verdict
        ok = int(st) == ost and (ost != 0 or abs(pr.objective - op.objective) <= 1e-5 * max(1.0, abs(op.objective)))
        if not ok:
            bad += 1
            print("MISMATCH", cfg, s, int(st), ost, pr.objective, op.objective, flush=True)
print("instances", n, "mismatches", bad, "%.0f s" % (time.time() - t0))

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
import planner_miqp_amd as P, oracle_lib
from planner_miqp_amd import synthetic
O = oracle_lib.Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
ps = [synthetic.generate("mini", s, gap=1e-6, max_time=60) for s in range(12)]
orc = []
for p in ps:
    h = O.from_params(p, 10); st, r, pr = O.solve(h, O.dims(p), gap=1e-6); orc.append(pr.objective); O.free(h)
for rep in range(3):
    singles = []
    for p in ps:
        w = P.CplexWrapper(); w.resetParameters(p); w.callCplex(); pr = w.getSolutionProperties(); singles.append((pr.objective, pr.nodes, pr.status))
    ws = []
    for p in ps:
        w = P.CplexWrapper(); w.resetParameters(p); ws.append(w)
    P.solve_batch(ws)
    batch = [(w.getSolutionProperties().objective, w.getSolutionProperties().nodes, w.getSolutionProperties().status) for w in ws]
    for k in range(12):
        bad = abs(singles[k][0] - orc[k]) > 1e-5 * orc[k] or abs(batch[k][0] - orc[k]) > 1e-5 * orc[k]
        if bad or rep == 0:
            print(rep, k, "oracle %.6f single %.6f (%d nodes, st %d) batch %.6f (%d nodes, st %d) %s" % (orc[k], singles[k][0], singles[k][1], singles[k][2], batch[k][0], batch[k][1], batch[k][2], "<<<< MISMATCH" if bad else ""))

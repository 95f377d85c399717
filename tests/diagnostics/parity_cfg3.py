"""One-off: full-size cfg3 instances that the CPU oracle can finish, device vs oracle at gap 1e-3 (GPU only)."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
import planner_miqp_amd as P, oracle_lib
from planner_miqp_amd import synthetic
O = oracle_lib.Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
G = 1e-3
ps = [synthetic.generate("cfg3", s, gap=G, max_time=15) for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300, (int(sys.argv[1]) if len(sys.argv) > 1 else 300) + 128)]
ws = []
for p in ps:
    w = P.CplexWrapper(); w.resetParameters(p); ws.append(w)
sts = P.solve_batch(ws)
def orc(p):
    h = O.from_params(p, 10); r = O.solve(h, O.dims(p), gap=G, time_limit=15); O.free(h); return r
with ThreadPoolExecutor(64) as ex:
    res = list(ex.map(orc, ps))
n = bad = 0
for k, (w, st, (ost, r, op)) in enumerate(zip(ws, sts, res)):
    pr = w.getSolutionProperties()
    if ost != 0 or op.gap > G + 1e-9 or int(st) != 0 or pr.gap > G + 1e-9:
        continue
    n += 1
    lo = max(pr.best_bound, op.best_bound); 
    if not (abs(pr.objective - op.objective) <= 2 * G * max(1.0, abs(op.objective)) and pr.objective >= op.best_bound - 1e-6 * abs(op.objective) and op.objective >= pr.best_bound - 1e-6 * abs(pr.objective)):
        bad += 1; print("MISMATCH seed", (int(sys.argv[1]) if len(sys.argv) > 1 else 300) + k, pr.objective, pr.best_bound, op.objective, op.best_bound, flush=True)
print("both solved", n, "of", len(ps), "mismatches", bad)

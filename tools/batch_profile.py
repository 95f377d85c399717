"""per-instance view of one bench batch (cfg3, seeds of bench.py's step `step`): nodes, finish time, status of the
hardest instances and the cumulative share of B&B nodes - who sets the batch's wall time.  python tools/batch_profile.py [step]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic

step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = 256
ws = []
for k in range(B):
    w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", step * B + k, gap=0.01, max_time=10.0)); ws.append(w)
t = time.time(); sts = P.solve_batch(ws); dt = time.time() - t
rows = []
for k, (w, st) in enumerate(zip(ws, sts)):
    pr = w.getSolutionProperties()
    rows.append((int(pr.nodes), round(pr.time, 3), step * B + k, int(st), pr.status, round(pr.gap, 5)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("batch %.2f s, %d nodes, solved %d / %d; timing %s" % (dt, tot, sum(1 for r in rows if r[4] in (101, 102)), len(rows), json.dumps(ws[0].lastTiming())))
cum = 0
for r in rows[:int(os.environ.get('BP_TOP', '24'))]:
    cum += r[0]
    print("seed %4d nodes %9d (cum %.3f) time %7.3f st %d cplex-status %d gap %.5f" % (r[2], r[0], cum / tot, r[1], r[3], r[4], r[5]))
ts = np.array(sorted(r[1] for r in rows))
print("finish-time quantiles", {q: float(np.quantile(ts, q)) for q in (0.5, 0.8, 0.9, 0.95, 0.98, 0.99)})

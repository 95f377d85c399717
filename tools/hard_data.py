"""outcome of every instance of a bench-like stream (seeds a .. b-1 of cfg3, 1280 in flight, 10 s limit): python tools/hard_data.py a b out.json
(round 5: data for an admission order by predicted hardness - which instances take the limit, which are slow)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
a, b, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ws = []
for sd in range(a, b):
    w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", sd, gap=0.01, max_time=10.0)); ws.append(w)
P.prepare_batch(ws)
t = time.time(); sts = P.solve_batch(ws, inflight=1280, prepared=True); dt = time.time() - t
rows = []
for sd, w, st in zip(range(a, b), ws, sts):
    pr = w.getSolutionProperties(); rows.append(dict(seed=sd, st=int(st), status=int(pr.status), time=float(pr.time), nodes=int(pr.nodes)))
json.dump(dict(seconds=dt, last_admission=ws[0].lastAdmission() if hasattr(ws[0], "lastAdmission") else None, rows=rows), open(out, "w"))
print("stream of %d: %.2f s, %d proven, last admission %s" % (b - a, dt, sum(r["status"] in (101, 102) for r in rows), ws[0].lastAdmission() if hasattr(ws[0], "lastAdmission") else "?"))

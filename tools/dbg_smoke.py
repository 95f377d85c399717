"""smoke solve with per-kernel synchronisation (MIQP_DEBUG_SYNC=1 python tools/dbg_smoke.py [cfg] [n] [inflight])"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
infl = int(sys.argv[3]) if len(sys.argv) > 3 else None
ws = []
for s in range(n):
    w = P.CplexWrapper(); w.resetParameters(synthetic.generate(cfg, s, gap=0.01, max_time=10.0)); ws.append(w)
sts = P.solve_batch(ws, inflight=infl)
for w, st in zip(ws, sts):
    pr = w.getSolutionProperties()
    print(int(st), pr.status, round(pr.objective, 6), round(pr.gap, 5), pr.nodes, round(pr.time, 4))

"""the same queue solved twice in one process (the second call reuses the device context): solved, nodes and time of both
calls - any difference in the node count is state that a call left behind.  python tools/twice_check.py cfg Q inflight first_seed"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 256
infl = int(sys.argv[3]) if len(sys.argv) > 3 else 256
first = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
for rep in range(3):
    ws = []
    for s in range(Q):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate(cfg, first + s, gap=0.01, max_time=10.0)); ws.append(w)
    P.prepare_batch(ws); t = time.time(); sts = P.solve_batch(ws, inflight=infl, prepared=True); dt = time.time() - t
    nodes = sum(w.getSolutionProperties().nodes for w in ws)
    solved = sum(int(st) == 0 and w.getSolutionProperties().status in (101, 102) for w, st in zip(ws, sts))
    tm = ws[0].lastTiming()
    print("call %d: solved %d of %d, nodes %d, wall %.2f s, rounds loop %.2f s, context built %s" % (rep, solved, Q, nodes, dt, tm["solve_s"], tm["context_built"]))

"""queue of Q cfg3 instances drained with `inflight` in flight (miqp_solver_solve_stream): solved, wall time, throughput,
latency quantiles (GPU only).  python tools/stream_check.py Q inflight [first_seed] [time_limit] [cfg] [gap]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
infl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tl = float(sys.argv[4]) if len(sys.argv) > 4 else 10.0
cfg = sys.argv[5] if len(sys.argv) > 5 else "cfg3"
gap = float(sys.argv[6]) if len(sys.argv) > 6 else 0.01
ws = []
for s in range(Q):
    w = P.CplexWrapper(); w.resetParameters(synthetic.generate(cfg, first + s, gap=gap, max_time=tl)); ws.append(w)
P.prepare_batch(ws); t = time.time(); sts = P.solve_batch(ws, inflight=infl, prepared=True); dt = time.time() - t
lat = []; nodes = 0; unsolved = []; un_nodes = 0; un_t = []
for k, (w, st) in enumerate(zip(ws, sts)):
    pr = w.getSolutionProperties(); nodes += pr.nodes
    if int(st) == 0 and pr.status in (101, 102): lat.append(pr.time)
    else: unsolved.append((first + k, int(st), pr.status, round(pr.gap, 4) if pr.gap == pr.gap else None, int(pr.nodes))); un_nodes += pr.nodes
print(json.dumps(dict(Q=Q, inflight=infl, seconds=round(dt, 3), solved=len(lat), solves_per_s=round(len(lat) / dt, 2), nodes=int(nodes), nodes_per_solved=round(nodes / max(1, len(lat))),
                      latency={q: round(float(np.percentile(lat, q)), 4) for q in (50, 90, 95, 99, 100)}, timing=ws[0].lastTiming(), unsolved_count=len(unsolved), unsolved_nodes=int(un_nodes), unsolved=unsolved[:12])))

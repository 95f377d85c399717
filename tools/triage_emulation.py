"""what an admission order by measured hardness would buy on a bench-like stream, emulated through the public API (round 5):
one stream of seeds a..b-1 at 1280 in flight with the 10 s limit, against (phase 1) the same queue with a short per-instance budget, then
(phase 2) the instances it left unproven, restarted from scratch with their full limit in the order of their gap after phase 1.
    python tools/triage_emulation.py a b budget"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
a, b, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
L = P.load_library()
def mk(limit):
    ws = []
    for sd in range(a, b):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", sd, gap=0.01, max_time=limit)); ws.append(w)
    P.prepare_batch(ws); return ws
def ok(w, st): pr = w.getSolutionProperties(); return st == P.OptimizationStatus.SUCCESS and pr.status in (101, 102)
# warm-up (device context of this shape)
P.solve_batch(mk(10.0)[:2560], inflight=1280, prepared=True)
ws = mk(10.0)
t = time.time(); sts = P.solve_batch(ws, inflight=1280, prepared=True); d0 = time.time() - t
n0 = sum(ok(w, s) for w, s in zip(ws, sts))
print("one phase : %d instances, %.2f s, %d proven -> %.0f solves/s" % (b - a, d0, n0, n0 / d0), flush=True)
ws = mk(budget)
t = time.time(); sts = P.solve_batch(ws, inflight=1280, prepared=True); d1 = time.time() - t
done = [ok(w, s) for w, s in zip(ws, sts)]
rest = [(w, w.getSolutionProperties()) for w, o in zip(ws, done) if not o]
rest.sort(key=lambda x: -(10.0 if x[1].gap != x[1].gap else x[1].gap))
ws2 = [w for w, _ in rest]
for w in ws2: L.miqp_solver_override_settings(w._h, 10.0 - budget, 0.01)
t = time.time(); sts2 = P.solve_batch(ws2, inflight=1280, prepared=True); d2 = time.time() - t
n1 = sum(done) + sum(ok(w, s) for w, s in zip(ws2, sts2))
print("two phases: budget %.2f s: phase 1 %.2f s proves %d, phase 2 (%d instances by gap, restarted) %.2f s -> total %.2f s, %d proven -> %.0f solves/s" % (budget, d1, sum(done), len(ws2), d2, d1 + d2, n1, n1 / (d1 + d2)), flush=True)

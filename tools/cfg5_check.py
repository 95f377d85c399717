"""cfg5 (4 cars x 30 steps x 64 regions), seeds 0-15, 10 s limit: one at a time and all 16 in flight.  python tools/cfg5_check.py [single|crowd|both]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
what = sys.argv[1] if len(sys.argv) > 1 else "both"
if what in ("single", "both"):
    ok = 0; tt = 0.0; rows = []
    for s in range(16):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg5", s, gap=0.01, max_time=10.0))
        t = time.time(); st = w.callCplex(); dt = time.time() - t
        pr = w.getSolutionProperties(); ok += pr.status in (101, 102); tt += dt
        rows.append("%d:%d/%.2fs" % (s, pr.status, dt))
    print("one at a time: proven %d of 16 in %.1f s  %s" % (ok, tt, " ".join(rows)), flush=True)
if what in ("crowd", "both"):
    ws = []
    for s in range(16):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg5", s, gap=0.01, max_time=10.0)); ws.append(w)
    t = time.time(); sts = P.solve_batch(ws, inflight=16); dt = time.time() - t
    prs = [w.getSolutionProperties() for w in ws]
    print("16 in flight: proven %d of 16 in %.1f s, %d rounds, %d nodes; not proven: %s" % (sum(p.status in (101, 102) for p in prs), dt, ws[0].lastTiming()["ipm_launches"], ws[0].lastTiming()["nodes"],
          [(k, p.status, None if p.gap != p.gap else round(p.gap, 3)) for k, p in enumerate(prs) if p.status not in (101, 102)]), flush=True)

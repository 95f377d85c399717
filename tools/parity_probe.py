"""device against oracle on the seeds of test_full_size_parity_at_a_tight_gap: objective difference and region mismatches per seed
(python tools/parity_probe.py cfg3 700 748 1e-7)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
import oracle_lib
from concurrent.futures import ThreadPoolExecutor
cfg = sys.argv[1]; s0, s1 = int(sys.argv[2]), int(sys.argv[3]); gap = float(sys.argv[4])
O = oracle_lib.Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
ps = [synthetic.generate(cfg, s, gap=gap, max_time=60) for s in range(s0, s1)]
ws = []
for p in ps:
    w = P.CplexWrapper(); w.resetParameters(p); ws.append(w)
sts = P.solve_batch(ws)
def orc(p):
    h = O.from_params(p, 10); r = O.solve(h, O.dims(p), gap=gap, time_limit=20); O.free(h); return r
with ThreadPoolExecutor(os.cpu_count() or 8) as ex:
    res = list(ex.map(orc, ps))
for s, p, w, st, (ost, ores, op) in zip(range(s0, s1), ps, ws, sts, res):
    pr = w.getSolutionProperties()
    if ost != 0 or op.status not in (101, 102):
        print(s, "oracle unfinished", ost, op.status); continue
    r = w.getRawResults()
    ra, rb = r.active_region.argmax(-1), ores.active_region.argmax(-1)
    mism = [(int(c), int(i), int(ra[c, i]), int(rb[c, i]), float(r.vel_y[c, i]), float(ores.vel_y[c, i])) for c, i in np.argwhere(ra != rb)]
    print(s, "dev %.9f orc %.9f rel %.2e gap %.1e nodes %d" % (pr.objective, op.objective, (pr.objective - op.objective) / max(1, abs(op.objective)), pr.gap, pr.nodes), "MISMATCH" if mism else "", mism[:4])

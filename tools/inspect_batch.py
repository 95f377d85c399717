"""Per-instance outcome of one bench batch: final gap, nodes, status (diagnostic; GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tl = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ps = [synthetic.generate(cfg, first + s, gap=0.01, max_time=tl) for s in range(n)]
ws = []
for p in ps:
    w = P.CplexWrapper(); w.resetParameters(p); ws.append(w)
P.solve_batch(ws)
rows = []
for k, w in enumerate(ws):
    pr = w.getSolutionProperties()
    rows.append((pr.gap if pr.gap == pr.gap else 9.99, first + k, pr.nodes, pr.status, pr.objective, pr.best_bound))
rows.sort()
solved = sum(1 for r in rows if r[0] <= 0.01 + 1e-12)
print("solved", solved, "of", n)
for r in rows:
    if r[0] > 0.01 + 1e-12:
        print("seed %3d gap %.4f nodes %8d status %d obj %.4f bound %.4f" % (r[1], r[0], r[2], r[3], r[4], r[5]))

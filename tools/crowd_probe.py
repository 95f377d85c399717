"""two hard instances at the head (or the tail) of a crowd of easy ones: what they cost in a crowd.  python tools/crowd_probe.py head|tail [n_easy]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
where = sys.argv[1] if len(sys.argv) > 1 else "head"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1278
hard = [1008, 1059]
easy = [s for s in range(2000, 2000 + 3 * n) if s not in hard][:n]
seeds = hard + easy if where == "head" else easy + hard
ws = []
for s in seeds:
    w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", s, gap=0.01, max_time=30)); ws.append(w)
t = time.time(); sts = P.solve_batch(ws, inflight=len(ws)); dt = time.time() - t
tot = 0
for s, w in zip(seeds, ws):
    pr = w.getSolutionProperties(); tot += pr.nodes
    if s in hard: print("seed %d: status %d nodes %d time %.2f" % (s, pr.status, pr.nodes, pr.time))
print("%s of %d: %.2f s, %d nodes in all, %d rounds" % (where, len(ws), dt, tot, ws[0].lastTiming()["ipm_launches"]))

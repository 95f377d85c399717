"""round-4 exploration on the GPU: where the hard tail sits.  (a) cfg5 seeds one at a time at two time limits; (b) a cfg3 queue with a long
limit: nodes and time per instance, the hardest ones, the share of the work they take.  python tools/explore_r4.py [a|b|ab]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
what = sys.argv[1] if len(sys.argv) > 1 else "ab"
if "a" in what:
    for tl in (10.0, 40.0):
        rows = []
        for s in range(16):
            w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg5", s, gap=0.01, max_time=tl))
            t = time.time(); st = w.callCplex(); dt = time.time() - t
            pr = w.getSolutionProperties()
            rows.append((s, int(st), pr.status, None if pr.gap != pr.gap else round(pr.gap, 4), int(pr.nodes), round(dt, 2), None if pr.objective != pr.objective else round(pr.objective, 2), int(pr.NrIterations)))
        print("cfg5 limit %.0f s: proven %d of 16" % (tl, sum(1 for r in rows if r[2] in (101, 102))))
        for r in rows:
            print("   seed %2d st %d status %d gap %s nodes %d time %.2f obj %s it/node %.1f" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7] / max(1, r[4])))
        sys.stdout.flush()
if "b" in what:
    Q, infl, tl = 2048, 256, 40.0
    ws = []
    for s in range(Q):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate("cfg3", s, gap=0.01, max_time=tl)); ws.append(w)
    P.prepare_batch(ws); t = time.time(); sts = P.solve_batch(ws, inflight=infl, prepared=True); dt = time.time() - t
    rec = []
    for k, (w, st) in enumerate(zip(ws, sts)):
        pr = w.getSolutionProperties()
        rec.append((k, int(st), pr.status, pr.gap, int(pr.nodes), pr.time))
    nodes = np.array([r[4] for r in rec], float); tm = np.array([r[5] for r in rec])
    order = np.argsort(-nodes)
    print("cfg3 queue of %d at %d in flight, limit %.0f s: %.2f s, proven %d, nodes %.3g; instances over 10 s: %d; share of all nodes in the hardest 1%% / 2%% / 5%%: %.2f / %.2f / %.2f"
          % (Q, infl, tl, dt, sum(1 for r in rec if r[2] in (101, 102)), nodes.sum(), int((tm > 10).sum()), nodes[order[:Q // 100]].sum() / nodes.sum(), nodes[order[:Q // 50]].sum() / nodes.sum(), nodes[order[:Q // 20]].sum() / nodes.sum()))
    print("nodes per instance: median %d mean %d p90 %d p99 %d max %d" % (np.median(nodes), nodes.mean(), np.percentile(nodes, 90), np.percentile(nodes, 99), nodes.max()))
    for k in order[:30]:
        r = rec[k]
        print("   seed %4d status %d gap %.4f nodes %8d time %.2f" % (r[0], r[2], r[3], r[4], r[5]))

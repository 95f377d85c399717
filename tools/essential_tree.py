"""how much of a hard instance's tree is incumbent-finding: solve, then solve again with the first solve's solution as MIP start
(the tree that remains is what the bound needs).  python tools/essential_tree.py cfg seed [seed ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg = sys.argv[1]
for s in sys.argv[2:]:
    p = synthetic.generate(cfg, int(s), gap=0.01, max_time=float(os.environ.get("TL", "40")))
    w = P.CplexWrapper(); w.resetParameters(p)
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    pr = w.getSolutionProperties()
    w2 = P.CplexWrapper(); w2.resetParameters(p); w2.addRecedingHorizonWarmstart(w.getRawResults())
    t = time.time(); st2 = w2.callCplex(); dt2 = time.time() - t
    pr2 = w2.getSolutionProperties()
    print("%s seed %s: plain status %d obj %.3f nodes %d time %.2f | with its solution as MIP start: status %d obj %.3f nodes %d time %.2f" % (
        cfg, s, pr.status, pr.objective, pr.nodes, dt, pr2.status, pr2.objective, pr2.nodes, dt2), flush=True)

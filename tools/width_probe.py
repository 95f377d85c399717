"""node count of single hard instances as a function of the round width (GPU only): python tools/width_probe.py seed...
WIDTHS=64,256 to choose the widths"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
widths = [int(x) for x in os.environ.get("WIDTHS", "64,256,1024,4096,16384").split(",")]
for seed in [int(a) for a in sys.argv[1:]]:
    for npr in widths:
        p = synthetic.generate("cfg3", seed, gap=0.01, max_time=20.0)
        w = P.CplexWrapper(nodes_per_round=npr); w.resetParameters(p)
        t = time.time(); st = w.callCplex(); dt = time.time() - t
        pr = w.getSolutionProperties(); tm = w.lastTiming()
        print("seed %d width %5d: status %d nodes %8d rounds %5d time %.3f s gap %.4f obj %.3f" % (seed, npr, pr.status, pr.nodes, tm["ipm_launches"], dt, pr.gap, pr.objective), flush=True)

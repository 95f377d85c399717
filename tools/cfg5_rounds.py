"""round widths and outcomes of single cfg5 solves (MIQP_ROUND_LOG=1 MIQP_STATS=1 python tools/cfg5_rounds.py seed ...  2> err.log)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
for s in sys.argv[1:]:
    p = synthetic.generate("cfg5", int(s), gap=0.01, max_time=float(os.environ.get("TL", "10")))
    w = P.CplexWrapper(); w.resetParameters(p)
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    pr = w.getSolutionProperties(); tm = w.lastTiming()
    print("cfg5 seed %s: st %d status %d gap %.4f obj %.3f nodes %d it/node %.1f time %.2f s, ipm launches %d, ipm %.2f s (%.2f ms per round), solve %.2f s" % (
        s, int(st), pr.status, pr.gap, pr.objective, pr.nodes, pr.NrIterations / max(1, pr.nodes), dt, tm["ipm_launches"], tm["ipm_s"], 1e3 * tm["ipm_s"] / max(1, tm["ipm_launches"]), tm["solve_s"]), flush=True)
    sys.stderr.write("=== end of seed %s\n" % s); sys.stderr.flush()

"""repeats of one single solve in one process: the set of (objective, gap, nodes, iterations) seen (python tools/repeat_as.py seed gap n)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
seed = int(sys.argv[1]); gap = float(sys.argv[2]); n = int(sys.argv[3])
p = synthetic.generate("cfg3", seed, gap=gap, max_time=10)
seen = {}
for k in range(n):
    w = P.CplexWrapper(); w.resetParameters(p); assert int(w.callCplex()) == 0
    s = w.getSolutionProperties(); tm = w.lastTiming()
    key = (float(s.objective).hex(), int(s.nodes), int(s.NrIterations), tm["as_nodes"], tm["as_steps"], tm["as_unfinished"], tm["as_drops"], tm["as_rows_end"], tm["as_rows_parent"])
    seen.setdefault(key, []).append(k)
for k, v in seen.items():
    print(k, v)

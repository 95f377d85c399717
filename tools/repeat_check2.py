"""Reproducibility of single solves of synthetic instances: each of the first `n` seeds of a config is solved `reps` times;
prints the seeds whose (objective, gap, nodes) differ between repetitions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gap = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
w = P.CplexWrapper()
bad = 0
for s in range(n):
    p = synthetic.generate(cfg, s, gap=gap, max_time=10.0)
    seen = {}
    for r in range(reps):
        w.resetParameters(p); w.callCplex(); pr = w.getSolutionProperties()
        key = (float(pr.objective).hex(), float(pr.gap).hex(), int(pr.nodes))
        seen[key] = seen.get(key, 0) + 1
    if len(seen) > 1:
        bad += 1
        print("seed", s, seen)
print("%s: %d of %d seeds differ between repetitions" % (cfg, bad, n))

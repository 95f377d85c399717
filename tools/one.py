"""One synthetic instance with progress output (diagnostic; GPU only): python tools/one.py cfg3 11 20"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
from planner_miqp_amd import synthetic
cfg, seed, tl = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
p = synthetic.generate(cfg, seed, gap=0.01, max_time=tl)
w = P.CplexWrapper(verbose=int(os.environ.get("VERBOSE", "1")), max_open_nodes=int(os.environ.get("OPEN_CAP", "0"))); w.resetParameters(p); st = w.callCplex(); pr = w.getSolutionProperties()
print("status", st, "obj", pr.objective, "bound", pr.best_bound, "gap", pr.gap, "nodes", pr.nodes, "time", pr.time)
r = w.getRawResults()
import numpy as np
N = p.NumSteps
for c in range(p.NumCars):
    print("car", c, "x", np.round(np.array(r.pos_x)[c], 2).tolist())
    print("car", c, "y", np.round(np.array(r.pos_y)[c], 2).tolist())
    print("car", c, "vx", np.round(np.array(r.vel_x)[c], 2).tolist())
    print("car", c, "vy", np.round(np.array(r.vel_y)[c], 2).tolist())
    print("car", c, "region", np.array(r.active_region)[c].argmax(-1).tolist() if hasattr(r, "active_region") else "")
print("x_ref", np.round(np.array(p.x_ref), 2).tolist()); print("y_ref", np.round(np.array(p.y_ref), 2).tolist())

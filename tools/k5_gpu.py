"""The reference's 3-car fixture cplexmodel.dat (K5) on the device (diagnostic; GPU only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import planner_miqp_amd as P
from helpers import dat_path
tl = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
w = P.CplexWrapper("cplexmodel.mod", P.ParameterSource.DATFILE, 12, gap_override=0.1, verbose=1)
w.setParameterDatFileAbsolute(dat_path("cplexmodel.dat"))
import ctypes
t = time.time(); st = w.callCplex(); pr = w.getSolutionProperties()
print("status", st, "obj", pr.objective, "bound", pr.best_bound, "gap", pr.gap, "nodes", pr.nodes, "%.2fs" % (time.time() - t))

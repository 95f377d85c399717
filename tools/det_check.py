"""run-to-run determinism of one solve (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import planner_miqp_amd as P
from helpers import load_params, dat_path
p = load_params("cplexmodel_testcase.dat"); p.relative_mip_gap_tolerance = 1e-3
for k in range(4):
    w = P.CplexWrapper(); w.resetParameters(p); w.callCplex(); pr = w.getSolutionProperties()
    print("cpp", repr(pr.objective), repr(pr.best_bound), pr.nodes, pr.NrIterations)
w.writeDat("/tmp/rt.dat")
for k in range(3):
    w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE); w.setParameterDatFileAbsolute("/tmp/rt.dat"); w.callCplex(); pr = w.getSolutionProperties()
    print("dat", repr(pr.objective), repr(pr.best_bound), pr.nodes, pr.NrIterations)

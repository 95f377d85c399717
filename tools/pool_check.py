import sys; sys.path.insert(0,'.')
import planner_miqp_amd as P
from tests.helpers import dat_path
for gap in (0.1, 0.01, 1e-6):
    w = P.CplexWrapper("cplexmodel.mod", P.ParameterSource.DATFILE, 12, gap_override=gap); w.setParameterDatFileAbsolute(dat_path("cplexmodel_testcase.dat"))
    st = w.callCplex(); pr = w.getSolutionProperties(); print("gap", gap, "pool", pr.NrSolutionPool, "nodes", pr.nodes, "obj", pr.objective, "time", pr.time)

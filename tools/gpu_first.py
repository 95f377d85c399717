import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import planner_miqp_amd as P
GD = os.path.join(ROOT, "tests", "golden")
for f, gap in (("cplexmodel_testcase.dat", 0.01), ("test_sos.dat", 0.01)):
    w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE, gap_override=gap, verbose=2 if "-v" in sys.argv else 0)
    w.setParameterDatFileAbsolute(os.path.join(GD, "ref_data", f))
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    print(f, st, w.getSolutionProperties(), "wall %.3f" % dt, w.lastTiming())
    t = time.time(); st = w.callCplex(); dt = time.time() - t
    print(" again:", st, "obj", w.getSolutionProperties().objective, "wall %.3f" % dt, w.lastTiming())
g = json.load(open(os.path.join(GD, "k3_testcase.json")))
w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE)
w.setParameterDatFileAbsolute(os.path.join(GD, "ref_data", "cplexmodel_testcase.dat"))
fx = P.RawResults(1, 20, 32, 1, 1, 4)
for n in ["active_region", "region_change_not_allowed_x_positive", "region_change_not_allowed_y_positive",
          "region_change_not_allowed_x_negative", "region_change_not_allowed_y_negative", "region_change_not_allowed_combined",
          "notWithinEnvironmentRear", "notWithinEnvironmentFrontUbUb", "notWithinEnvironmentFrontLbUb",
          "notWithinEnvironmentFrontUbLb", "notWithinEnvironmentFrontLbLb", "deltacc", "deltacc_front"]:
    a = getattr(fx, n); a[...] = np.array(g[n], dtype=np.int32).reshape(a.shape)
rc, out, obj, it = w.solveFixed(fx)
print("solveFixed rc", rc, "obj", obj, "it", it)
for n in ["pos_x", "vel_x", "pos_y", "vel_y", "acc_y", "u_y", "pos_x_front_UB", "pos_y_front_UB", "pos_y_front_LB"]:
    print("  ", n, np.abs(getattr(out, n).reshape(-1) - np.array(g[n])).max())

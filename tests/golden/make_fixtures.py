"""Regenerates the committed fixtures from the reference checkout (run in the build container only).

  * tests/golden/ref_data/*.dat, modelRun.txt : DATA files the reference's own tests hold
    (cplexmodel/cplexmodel_testcase.dat, test_sos.dat, cplexmodel.dat, modelRun.txt), copied verbatim.
  * tests/golden/k3_testcase.json : the known answers K1-K3 of test/cplex_wrapper_test.cc
    (:283-456 solution vector, :866-874 sizes and objective), transcribed as numbers.
/root/reference does not exist on the GPU box; tests read only the committed copies.
"""
import json
import os
import re
import shutil
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_setvalues(src, name):
    """numbers of `wv.<name>.setValues({...});` in the gtest source, flattened"""
    m = re.search(r"wv\." + name + r"\.setValues\(\s*(\{.*?\})\s*\)\s*;", src, re.S)
    body = m.group(1)
    return [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)]


def main():
    os.makedirs(os.path.join(HERE, "ref_data"), exist_ok=True)
    for f in ["cplexmodel_testcase.dat", "test_sos.dat", "cplexmodel.dat", "modelRun.txt"]:
        shutil.copyfile(os.path.join(REF, "cplexmodel", f), os.path.join(HERE, "ref_data", f))
    src = open(os.path.join(REF, "test", "cplex_wrapper_test.cc")).read()
    names = ["pos_x", "vel_x", "acc_x", "pos_y", "vel_y", "acc_y", "u_x", "u_y", "pos_x_front_UB", "pos_x_front_LB",
             "pos_y_front_UB", "pos_y_front_LB", "active_region", "region_change_not_allowed_x_positive",
             "region_change_not_allowed_y_positive", "region_change_not_allowed_x_negative",
             "region_change_not_allowed_y_negative", "region_change_not_allowed_combined", "notWithinEnvironmentRear",
             "notWithinEnvironmentFrontUbUb", "notWithinEnvironmentFrontLbUb", "notWithinEnvironmentFrontUbLb",
             "notWithinEnvironmentFrontLbLb", "deltacc", "deltacc_front"]
    g = {n: parse_setvalues(src, n) for n in names}
    g["sizes"] = dict(NrConstraints=12361, NrBinaryVariables=1240, NrFloatVariables=340, NonZeroCoefficients=29834)
    g["objective"] = 9.57603
    g["objective_tol"] = 1e-5
    g["source"] = "test/cplex_wrapper_test.cc:283-456,866-874"
    json.dump(g, open(os.path.join(HERE, "k3_testcase.json"), "w"), indent=0)
    print("fixtures written", {k: len(v) for k, v in g.items() if isinstance(v, list)})




def region_tables():
    """planner_miqp_amd/data/region_tables_{16,32}.json: the fitted region tables (numeric data of
    common/parameter/fitting_polynomial_parameters.hpp as they appear in the reference's .dat fixtures:
    cplexmodel_testcase.dat = (32 regions, vmax 20), test_sos.dat = (16 regions, vmax 20)) used by the
    synthetic instance generator."""
    sys.path.insert(0, os.path.join(HERE, "..", "..", "tools"))
    from miqp_py.dat import load_dat
    out_dir = os.path.join(HERE, "..", "..", "planner_miqp_amd", "data")
    os.makedirs(out_dir, exist_ok=True)
    for f, R in (("cplexmodel_testcase.dat", 32), ("test_sos.dat", 16)):
        d = load_dat(os.path.join(REF, "cplexmodel", f))
        keys = ["fraction_parameters", "POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX",
                "POLY_KAPPA_AX_MIN", "min_acc_x", "max_acc_x", "min_acc_y", "max_acc_y", "min_jerk_x", "max_jerk_x",
                "min_jerk_y", "max_jerk_y", "total_min_acc", "total_max_acc", "total_min_jerk", "total_max_jerk",
                "min_vel_x_y", "max_vel_x_y", "minimum_region_change_speed"]
        t = {k: d[k] for k in keys}
        t["nr_regions"] = R
        t["source"] = "cplexmodel/" + f
        json.dump(t, open(os.path.join(out_dir, "region_tables_%d.json" % R), "w"))


def region_tables_64():
    """planner_miqp_amd/data/region_tables_64.json for the 4-car config cfg5 (BASELINE.json: 64 regions).  Sources:
    fraction parameters and acceleration / jerk boxes computed by the planner core (ParameterPreparer restatement, fitting
    speed 10 m/s, straight-line limits 2 / -4 / 3 / 1.6 / 1.4 as in src/miqp_planner_data.hpp:190-242); front-axle
    polynomials from the reference's DATA file data/polynoms_from_fitting_64_using_theta.mat (the constant-heading
    variant, the only 64-region fit shipped as data); the curvature polynomials exist for 64 regions only as C++ literals
    (common/parameter/fitting_polynomial_parameters.hpp), which are not copied: every 64-sector takes the fitted curvature
    polynomial of the 32-sector that contains it - an approximation, good enough for a synthetic benchmark instance."""
    import numpy as np
    import scipy.io as sio
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import planner_miqp_amd as P
    from planner_miqp_amd import planner_core as K
    P.build_library()
    R = 64
    out_dir = os.path.join(HERE, "..", "..", "planner_miqp_amd", "data")
    pp = K.ParameterPreparer(R, 10, 1, 2, -4, 3, 1.6, 1.4)
    acc, jerk = pp.CalculateAccLimitsPerCar(), pp.CalculateJerkLimitsPerCar()
    m = sio.loadmat(os.path.join(REF, "data", "polynoms_from_fitting_64_using_theta.mat"), squeeze_me=True, struct_as_record=False)["lin_result"]
    t32 = json.load(open(os.path.join(out_dir, "region_tables_32.json")))
    k32max, k32min = np.array(t32["POLY_KAPPA_AX_MAX"]).reshape(32, 3), np.array(t32["POLY_KAPPA_AX_MIN"]).reshape(32, 3)
    t = {"fraction_parameters": pp.GetFractionParameters().round(10).tolist(),
         "POLY_SINT_UB": np.array([m.poly_sin_ub[k] for k in range(R)]).round(10).tolist(),
         "POLY_SINT_LB": np.array([m.poly_sin_lb[k] for k in range(R)]).round(10).tolist(),
         "POLY_COSS_UB": np.array([m.poly_cos_ub[k] for k in range(R)]).round(10).tolist(),
         "POLY_COSS_LB": np.array([m.poly_cos_lb[k] for k in range(R)]).round(10).tolist(),
         "POLY_KAPPA_AX_MAX": k32max[np.arange(R) // 2].tolist(), "POLY_KAPPA_AX_MIN": k32min[np.arange(R) // 2].tolist()}
    for nm, d in (("acc", acc), ("jerk", jerk)):
        for k in ("min_x", "max_x", "min_y", "max_y"):
            t["%s_%s_%s" % (k[:3], nm, k[-1])] = [np.round(d[k], 4).tolist()]
    t["total_min_acc"] = float(min(min(t["min_acc_x"][0]), min(t["min_acc_y"][0]))); t["total_max_acc"] = float(max(max(t["max_acc_x"][0]), max(t["max_acc_y"][0])))
    t["total_min_jerk"] = float(min(min(t["min_jerk_x"][0]), min(t["min_jerk_y"][0]))); t["total_max_jerk"] = float(max(max(t["max_jerk_x"][0]), max(t["max_jerk_y"][0])))
    t.update(min_vel_x_y=-10, max_vel_x_y=10, minimum_region_change_speed=1, nr_regions=R,
             source="tests/golden/make_fixtures.py::region_tables_64 (planner core + data/polynoms_from_fitting_64_using_theta.mat + 32-region curvature fit)")
    json.dump(t, open(os.path.join(out_dir, "region_tables_64.json"), "w"))


if __name__ == "__main__":
    main()
    region_tables()
    region_tables_64()

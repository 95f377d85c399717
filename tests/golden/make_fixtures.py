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


def fitting_tables():
    """planner_miqp_amd/data/fitting_polynomial_parameters.json and planner_miqp_amd/csrc/fitting_tables.inc: the fitted
    polynomial tables of all seven (nr_regions, max_velocity_fitting, min_velocity_fitting) variants the reference ships
    (common/parameter/fitting_polynomial_parameters.hpp:28-92 map, :195-1281 numbers; outputs of the offline MATLAB fit).
    Only the NUMBERS are taken (the same treatment as K3), re-laid out row-major [region][coefficient]; the reference keeps
    them column-major (Eigen::Map of R x 3, :97-168)."""
    import numpy as np
    src = open(os.path.join(REF, "common", "parameter", "fitting_polynomial_parameters.hpp")).read()
    vecs = {}
    for m in re.finditer(r"const std::vector<double> (POLY_[A-Z_0-9]+) = \{(.*?)\};", src, re.S):
        vecs[m.group(1)] = [float(x) for x in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?", m.group(2))]
    assign = re.findall(r"(POLY_[A-Z_]+)_map\[\{(\d+), (\d+), (\d+)\}\] = (POLY_[A-Z_0-9]+);", src)
    combos = {}
    for kind, R, vmax, vmin, name in assign:
        R = int(R); v = vecs[name]
        assert len(v) == 3 * R, (name, len(v))
        combos.setdefault((R, int(vmax), int(vmin)), {})[kind] = np.array(v).reshape(3, R).T.tolist()   # column-major R x 3 -> [R][3]
    kinds = ["POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX", "POLY_KAPPA_AX_MIN"]
    assert len(combos) == 7 and all(sorted(c) == sorted(kinds) for c in combos.values())
    out_dir = os.path.join(HERE, "..", "..", "planner_miqp_amd", "data")
    js = {"source": "common/parameter/fitting_polynomial_parameters.hpp:28-92,195-1281 (numbers only, row-major [region][3])",
          "variants": [dict(nr_regions=k[0], max_velocity_fitting=k[1], min_velocity_fitting=k[2], **combos[k]) for k in sorted(combos)]}
    json.dump(js, open(os.path.join(out_dir, "fitting_polynomial_parameters.json"), "w"))
    with open(os.path.join(HERE, "..", "..", "planner_miqp_amd", "csrc", "fitting_tables.inc"), "w") as f:
        f.write("// GENERATED by tests/golden/make_fixtures.py::fitting_tables - numeric tables only (offline least-squares fits shipped\n"
                "// with the reference, common/parameter/fitting_polynomial_parameters.hpp), row-major [region][3];\n"
                "// order of the six tables: SINT_UB, SINT_LB, COSS_UB, COSS_LB, KAPPA_AX_MAX, KAPPA_AX_MIN\n")
        for k in sorted(combos):
            f.write("static const double FIT_%d_%d_%d[6][%d] = {\n" % (k[0], k[1], k[2], 3 * k[0]))
            for kind in kinds:
                f.write("  {" + ", ".join(repr(float(x)) for row in combos[k][kind] for x in row) + "},\n")
            f.write("};\n")
        f.write("static const struct { int R, vmax, vmin; const double (*t)[1]; } FIT_VARIANTS[] = {\n")
        for k in sorted(combos):
            f.write("  {%d, %d, %d, (const double (*)[1])FIT_%d_%d_%d},\n" % (k + k))
        f.write("};\n")
    print("fitting tables:", sorted(combos))


def region_tables_64():
    """planner_miqp_amd/data/region_tables_64.json for the 4-car config cfg5 (BASELINE.json: 64 regions).  Sources:
    fraction parameters and acceleration / jerk boxes computed by the planner core (ParameterPreparer restatement, fitting
    speed 10 m/s, straight-line limits 2 / -4 / 3 / 1.6 / 1.4 as in src/miqp_planner_data.hpp:190-242); front-axle
    and curvature polynomials = the reference's (64, 10, 1) fit, the numbers of
    common/parameter/fitting_polynomial_parameters.hpp:875-1281 as extracted by fitting_tables() above."""
    import numpy as np
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import planner_miqp_amd as P
    from planner_miqp_amd import planner_core as K
    P.build_library()
    R = 64
    out_dir = os.path.join(HERE, "..", "..", "planner_miqp_amd", "data")
    pp = K.ParameterPreparer(R, 10, 1, 2, -4, 3, 1.6, 1.4)
    acc, jerk = pp.CalculateAccLimitsPerCar(), pp.CalculateJerkLimitsPerCar()
    fit = json.load(open(os.path.join(out_dir, "fitting_polynomial_parameters.json")))
    v = [x for x in fit["variants"] if (x["nr_regions"], x["max_velocity_fitting"], x["min_velocity_fitting"]) == (64, 10, 1)][0]
    t = {"fraction_parameters": pp.GetFractionParameters().round(10).tolist()}
    for k in ("POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX", "POLY_KAPPA_AX_MIN"):
        t[k] = v[k]
    for nm, d in (("acc", acc), ("jerk", jerk)):
        for k in ("min_x", "max_x", "min_y", "max_y"):
            t["%s_%s_%s" % (k[:3], nm, k[-1])] = [np.round(d[k], 4).tolist()]
    t["total_min_acc"] = float(min(min(t["min_acc_x"][0]), min(t["min_acc_y"][0]))); t["total_max_acc"] = float(max(max(t["max_acc_x"][0]), max(t["max_acc_y"][0])))
    t["total_min_jerk"] = float(min(min(t["min_jerk_x"][0]), min(t["min_jerk_y"][0]))); t["total_max_jerk"] = float(max(max(t["max_jerk_x"][0]), max(t["max_jerk_y"][0])))
    t.update(min_vel_x_y=-10, max_vel_x_y=10, minimum_region_change_speed=1, nr_regions=R,
             source="tests/golden/make_fixtures.py::region_tables_64 (planner core + the (64, 10, 1) fit of fitting_polynomial_parameters.json)")
    json.dump(t, open(os.path.join(out_dir, "region_tables_64.json"), "w"))


if __name__ == "__main__":
    main()
    region_tables()
    fitting_tables()
    region_tables_64()

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


if __name__ == "__main__":
    sys.exit(main())

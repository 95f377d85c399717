"""shared helpers of the test-suite"""
import json
import os

import numpy as np

from planner_miqp_amd.ctypes_types import ModelParameters, RawResults

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BIN_FIELDS = ["active_region", "region_change_not_allowed_x_positive", "region_change_not_allowed_y_positive",
              "region_change_not_allowed_x_negative", "region_change_not_allowed_y_negative",
              "region_change_not_allowed_combined", "notWithinEnvironmentRear", "notWithinEnvironmentFrontUbUb",
              "notWithinEnvironmentFrontLbUb", "notWithinEnvironmentFrontUbLb", "notWithinEnvironmentFrontLbLb", "deltacc",
              "deltacc_front"]
CONT_FIELDS = ["u_x", "u_y", "pos_x", "vel_x", "acc_x", "pos_y", "vel_y", "acc_y", "pos_x_front_UB", "pos_x_front_LB",
               "pos_y_front_UB", "pos_y_front_LB"]


def dat_path(name):
    return os.path.join(GOLDEN, "ref_data", name)


def load_params(name):
    from miqp_py.dat import load_dat
    return ModelParameters.from_dat_dict(load_dat(dat_path(name)))


def k3():
    return json.load(open(os.path.join(GOLDEN, "k3_testcase.json")))


def k3_results():
    """K3 as a RawResults record (test/cplex_wrapper_test.cc:283-456)"""
    g = k3()
    r = RawResults(1, 20, 32, 1, 1, 4)
    for n in BIN_FIELDS:
        a = getattr(r, n)
        a[...] = np.array(g[n], dtype=np.int32).reshape(a.shape)
    for n in CONT_FIELDS:
        a = getattr(r, n)
        a[...] = np.array(g[n], dtype=np.float64).reshape(a.shape)
    for n in ["slackvarsObstacle", "slackvarsObstacle_front"]:
        getattr(r, n)[...] = 0
    return r, g

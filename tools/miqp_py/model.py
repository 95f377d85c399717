"""Independent numpy restatement of the planner-miqp MIQP (cplexmodel/*.mod).

This is a CHECKER (used by tests/ and by tests/golden/gen_golden.py), not the
product: it builds the model in *disjunctive* form - every group of big-M
binaries of the OPL model is one disjunction whose alternatives are sets of
plain linear rows over the stage vector z_i = [x_i(6C) | u_i(2C)] - and solves
node relaxations with a dense condensed primal-dual interior point method.

Reference files restated (paths relative to /root/reference):
  cplexmodel/parameters.mod:24-32          big-M constants
  cplexmodel/initialization.mod:13-32      RR, initial heading
  cplexmodel/objective_function.mod:7-19   objective
  cplexmodel/initial_conditions.mod:8-61   A1
  cplexmodel/model_region_constraints.mod:11-117  A2-A4
  cplexmodel/minimum_speed_constraints.mod:9-49   A5
  cplexmodel/obstacle_environment_constraints.mod:6-109  A6, A7
  cplexmodel/agent_collision_constraints.mod:10-73       A8
"""
import math
import numpy as np

BIGM = dict(jerk=10.0, velfrac=1000.0, pospoly=100.0, acc=10.0, kappa=1000.0,
            vel=100.0, env=10000.0, obs=10000.0, agents=1000.0)

PX, VX, AX, PY, VY, AY = range(6)

# point types: index into the per-car table of affine point coordinates
PT_R, PT_U, PT_L = 0, 1, 2
# env corner order (obstacle_environment_constraints.mod:15-28):
#   Rear, FrontUbUb, FrontLbUb, FrontUbLb, FrontLbLb ; name = (x-type)(y-type)
ENV_PTS = [(PT_R, PT_R), (PT_U, PT_U), (PT_L, PT_U), (PT_U, PT_L), (PT_L, PT_L)]
# obstacle corner order (:64-68): rear, (LB,LB), (UB,LB), (LB,UB), (UB,UB)
OBS_PTS = [(PT_R, PT_R), (PT_L, PT_L), (PT_U, PT_L), (PT_L, PT_U), (PT_U, PT_U)]


def round_p(v, p=10):
    """model_input_data_source.hpp:74-79 RoundWithPrecision."""
    s = 10.0 ** p
    return np.round(np.asarray(v, dtype=float) * s) / s


class Inst:
    """Numeric instance = ModelParameters (src/miqp_planner_data.hpp:99-185)."""

    @staticmethod
    def from_dat(d):
        I = Inst()
        I.N = int(d["NumSteps"]); I.C = int(d["NumCars"]); I.R = int(d["nr_regions"])
        I.ts = float(d["ts"])
        I.vmin = float(d["min_vel_x_y"]); I.vmax = float(d["max_vel_x_y"])
        I.amin = float(d["total_min_acc"]); I.amax = float(d["total_max_acc"])
        I.jmin = float(d["total_min_jerk"]); I.jmax = float(d["total_max_jerk"])
        I.safety = np.array(d["agent_safety_distance"], float)
        I.safety_slack = np.array(d["agent_safety_distance_slack"], float)
        I.max_slack = float(d["maximum_slack"])
        C, N, R = I.C, I.N, I.R
        I.W = np.zeros((C, 8))  # px vx ax py vy ay jx jy
        for k, nm in enumerate(["POS_X", "VEL_X", "ACC_X", "POS_Y", "VEL_Y", "ACC_Y", "JERK_X", "JERK_Y"]):
            I.W[:, k] = np.array(d["WEIGHTS_" + nm], float)
        I.w_slack = float(d["WEIGHTS_SLACK"]); I.w_slack_obs = float(d["WEIGHTS_SLACK_OBSTACLE"])
        I.wb = np.array(d["WheelBase"], float); I.rad = np.array(d["CollisionRadius"], float)
        I.x0 = np.array(d["IntitialState"], float).reshape(C, 6)
        I.ref = np.zeros((C, N, 6))
        I.ref[:, :, PX] = np.array(d["x_ref"], float).reshape(C, N)
        I.ref[:, :, VX] = np.array(d["vx_ref"], float).reshape(C, N)
        I.ref[:, :, PY] = np.array(d["y_ref"], float).reshape(C, N)
        I.ref[:, :, VY] = np.array(d["vy_ref"], float).reshape(C, N)

        def tab(name, width):
            a = np.array(d[name], float)
            a = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a.reshape(1, -1)
            return a

        def car_region(name):
            a = np.array(d[name], float).reshape(C, -1)
            out = np.zeros((C, R)); out[:, :a.shape[1]] = a  # cplexmodel.dat: 16-wide tables, R=32 header
            return out
        I.acc_lim = np.stack([car_region("min_acc_x"), car_region("max_acc_x"),
                              car_region("min_acc_y"), car_region("max_acc_y")], -1)
        I.jerk_lim = np.stack([car_region("min_jerk_x"), car_region("max_jerk_x"),
                               car_region("min_jerk_y"), car_region("max_jerk_y")], -1)
        I.init_region = np.array(d["initial_region"], int).reshape(C)  # 1-based
        pr = np.array(d["possible_region"], int).reshape(C, -1)
        I.possible = np.zeros((C, R), int); I.possible[:, :pr.shape[1]] = pr

        def reg_tab(name, w):
            a = np.array(d[name], float).reshape(-1, w)
            out = np.zeros((R, w)); out[:a.shape[0]] = a
            return out
        I.frac = reg_tab("fraction_parameters", 4)
        I.vm = float(d["minimum_region_change_speed"])
        I.poly = {k: reg_tab("POLY_" + k, 3) for k in
                  ["SINT_UB", "SINT_LB", "COSS_UB", "COSS_LB", "KAPPA_AX_MAX", "KAPPA_AX_MIN"]}
        # polygons -> edge arrays [x1,y1,x2,y2]
        I.env = [np.array([t[1:] for t in e], float) for e in d["MultiEnvironmentConvexPolygon"]]
        I.E = len(I.env)
        I.obs = [[np.array([t[1:] for t in poly], float) for poly in o] for o in d["ObstacleConvexPolygon"]]
        I.O = len(I.obs)
        I.obs_soft = [int(v) for v in d["obstacle_is_soft"]] if I.O else []
        I.L = int(d["max_lines_obstacles"])
        I.gap = float(d["relative_mip_gap_tolerance"]); I.tilim = float(d["max_solution_time"])
        return I

    # ---- derived -------------------------------------------------------
    def A_B(self):
        ts = self.ts
        A1 = np.array([[1, ts, ts * ts / 2], [0, 1, ts], [0, 0, 1.0]])
        B1 = np.array([ts ** 3 / 6, ts * ts / 2, ts])
        C = self.C
        A = np.zeros((6 * C, 6 * C)); B = np.zeros((6 * C, 2 * C))
        for c in range(C):
            for ax in range(2):
                o = 6 * c + 3 * ax
                A[o:o + 3, o:o + 3] = A1
                B[o:o + 3, 2 * c + ax] = B1
        return A, B

    def theta0(self, c):
        return math.atan2(self.x0[c, VY], self.x0[c, VX])


def golden_k3():
    """test/cplex_wrapper_test.cc:283-456 (K3) - continuous part + regions."""
    g = {}
    g["pos_x"] = list(range(20))
    g["vel_x"] = [5] * 20
    g["pos_y"] = [0, 0.019179, 0.033586, 0.039222, 0.032977, 0.012594, -0.023365, -0.075549, -0.14385, -0.22744,
                  -0.3248, -0.4338, -0.55177, -0.67568, -0.80226, -0.92832, -1.0511, -1.1688, -1.2806, -1.3862]
    g["vel_y"] = [0.1, 0.087683, 0.053065, 0.00070665, -0.065036, -0.13999, -0.22014, -0.30162, -0.38071, -0.4539,
                  -0.5179, -0.56983, -0.60736, -0.62894, -0.63416, -0.62411, -0.60239, -0.57419, -0.54347, -0.51209]
    g["acc_y"] = [0, -0.12317, -0.22301, -0.30057, -0.35685, -0.39267, -0.40881, -0.40598, -0.38498, -0.34687,
                  -0.29317, -0.22615, -0.1491, -0.066723, 0.014515, 0.085985, 0.13124, 0.15069, 0.15655, 0.15729]
    g["u_y"] = [-0.61586, -0.49917, -0.38784, -0.28136, -0.17914, -0.080684, 0.014167, 0.10499, 0.19055, 0.26848,
                0.33512, 0.38525, 0.41188, 0.40619, 0.35735, 0.22628, 0.097256, 0.029291, 0.0036965, 0]
    g["pos_x_front_UB"] = [2.7994, 3.8165, 4.817, 5.8178, 6.8166, 7.8155, 8.8142, 9.813, 10.812, 11.811, 12.81,
                           13.809, 14.808, 15.808, 16.808, 17.808, 18.808, 19.809, 20.809, 21.81]
    g["pos_x_front_LB"] = [2.7994, 3.7458, 4.7463, 5.7472, 6.7462, 7.745, 8.7438, 9.7425, 10.741, 11.74, 12.739,
                           13.738, 14.738, 15.737, 16.737, 17.737, 18.738, 19.738, 20.739, 21.739]
    g["pos_y_front_UB"] = [0.055989, 0.43241, 0.44113, 0.43816, 0.038049, 0.0073762, -0.039585, -0.10296, -0.18212,
                           -0.27575, -0.3819, -0.49802, -0.62115, -0.74802, -0.87531, -1, -1.1198, -1.2337, -1.3412,
                           -1.4425]
    g["pos_y_front_LB"] = [0.055989, 0.017216, 0.026871, 0.025319, -0.38145, -0.41375, -0.46246, -0.5276, -0.60849,
                           -0.70371, -0.81125, -0.92851, -1.0525, -1.1798, -1.3072, -1.4317, -1.551, -1.6643,
                           -1.7712, -1.8717]
    g["region"] = [1, 1, 1, 1] + [32] * 16  # 1-based active region per step
    g["objective"] = 9.57603
    return g

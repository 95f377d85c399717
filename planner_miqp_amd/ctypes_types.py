"""ctypes mirrors of include/miqp_types.h (POD contract of the solve path).

ModelParameters / RawResults here are the Python-side records with the reference's member names
(src/miqp_planner_data.hpp:46-185); ``to_c`` / ``alloc`` produce the flat row-major C structs.
"""
import ctypes as C

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class ModelParamsC(C.Structure):
    _fields_ = [
        ("max_solution_time", C.c_double), ("relative_mip_gap_tolerance", C.c_double),
        ("mipdisplay", C.c_int), ("mipemphasis", C.c_int), ("relobjdif", C.c_double),
        ("cutpass", C.c_int), ("probe", C.c_int), ("repairtries", C.c_int), ("rinsheur", C.c_int),
        ("varsel", C.c_int), ("mircuts", C.c_int), ("parallelmode", C.c_int),
        ("NumSteps", C.c_int), ("nr_regions", C.c_int), ("NumCars", C.c_int), ("nr_obstacles", C.c_int),
        ("max_lines_obstacles", C.c_int), ("nr_environments", C.c_int),
        ("ts", C.c_double), ("min_vel_x_y", C.c_double), ("max_vel_x_y", C.c_double),
        ("total_min_acc", C.c_double), ("total_max_acc", C.c_double), ("total_min_jerk", C.c_double),
        ("total_max_jerk", C.c_double), ("maximum_slack", C.c_double), ("WEIGHTS_SLACK", C.c_double),
        ("WEIGHTS_SLACK_OBSTACLE", C.c_double), ("minimum_region_change_speed", C.c_double),
        ("agent_safety_distance", c_double_p), ("agent_safety_distance_slack", c_double_p),
        ("WEIGHTS_POS_X", c_double_p), ("WEIGHTS_VEL_X", c_double_p), ("WEIGHTS_ACC_X", c_double_p),
        ("WEIGHTS_POS_Y", c_double_p), ("WEIGHTS_VEL_Y", c_double_p), ("WEIGHTS_ACC_Y", c_double_p),
        ("WEIGHTS_JERK_X", c_double_p), ("WEIGHTS_JERK_Y", c_double_p),
        ("WheelBase", c_double_p), ("CollisionRadius", c_double_p), ("IntitialState", c_double_p),
        ("x_ref", c_double_p), ("vx_ref", c_double_p), ("y_ref", c_double_p), ("vy_ref", c_double_p),
        ("min_acc_x", c_double_p), ("max_acc_x", c_double_p), ("min_acc_y", c_double_p), ("max_acc_y", c_double_p),
        ("min_jerk_x", c_double_p), ("max_jerk_x", c_double_p), ("min_jerk_y", c_double_p), ("max_jerk_y", c_double_p),
        ("initial_region", c_int_p), ("possible_region", c_int_p),
        ("obstacle_vertices", c_double_p), ("obstacle_is_soft", c_int_p),
        ("env_offsets", c_int_p), ("env_vertices", c_double_p),
        ("fraction_parameters", c_double_p),
        ("POLY_SINT_UB", c_double_p), ("POLY_SINT_LB", c_double_p), ("POLY_COSS_UB", c_double_p),
        ("POLY_COSS_LB", c_double_p), ("POLY_KAPPA_AX_MAX", c_double_p), ("POLY_KAPPA_AX_MIN", c_double_p),
    ]


_RES_D = ["u_x", "u_y", "pos_x", "vel_x", "acc_x", "pos_y", "vel_y", "acc_y",
          "pos_x_front_UB", "pos_x_front_LB", "pos_y_front_UB", "pos_y_front_LB"]
_RES_I = ["notWithinEnvironmentRear", "notWithinEnvironmentFrontUbUb", "notWithinEnvironmentFrontLbUb",
          "notWithinEnvironmentFrontUbLb", "notWithinEnvironmentFrontLbLb", "active_region",
          "region_change_not_allowed_x_positive", "region_change_not_allowed_y_positive",
          "region_change_not_allowed_x_negative", "region_change_not_allowed_y_negative",
          "region_change_not_allowed_combined", "deltacc", "deltacc_front", "car2car_collision", "slackvars",
          "slackvarsObstacle", "slackvarsObstacle_front"]


class RawResultsC(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ["N", "NrEnvironments", "NrRegions", "NrObstacles", "MaxLinesObstacles",
                                        "NrCarToCarCollisions", "NrCars"]]
                + [(n, c_double_p) for n in _RES_D] + [(n, c_int_p) for n in _RES_I]
                + [("slackvars_real", c_double_p)])


class SolutionPropertiesC(C.Structure):
    _fields_ = [("status", C.c_int), ("gap", C.c_double), ("objective", C.c_double), ("time", C.c_double),
                ("NrConstraints", C.c_int), ("NrBinaryVariables", C.c_int), ("NrFloatVariables", C.c_int),
                ("NonZeroCoefficients", C.c_int), ("NrIterations", C.c_int), ("NrSolutionPool", C.c_int),
                ("best_bound", C.c_double), ("nodes", C.c_longlong)]


class SolverOptsC(C.Structure):
    _fields_ = [("precision", C.c_int), ("device", C.c_int), ("nodes_per_round", C.c_int),
                ("max_open_nodes", C.c_int), ("gap_override", C.c_double), ("verbose", C.c_int)]


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


class ModelParameters:
    """Python record with the member names of the reference's ModelParameters
    (src/miqp_planner_data.hpp:99-185).  Region limit tables are stored as min/max_acc/jerk_x/y [C,R];
    polygons as lists of CCW vertex arrays: MultiEnvironmentConvexPolygon[e] -> [n,2],
    ObstacleConvexPolygon[o][i] -> [L,2]."""

    SCALARS_F = ["max_solution_time", "relative_mip_gap_tolerance", "relobjdif", "ts", "min_vel_x_y", "max_vel_x_y",
                 "total_min_acc", "total_max_acc", "total_min_jerk", "total_max_jerk", "maximum_slack",
                 "WEIGHTS_SLACK", "WEIGHTS_SLACK_OBSTACLE", "minimum_region_change_speed"]
    SCALARS_I = ["mipdisplay", "mipemphasis", "cutpass", "probe", "repairtries", "rinsheur", "varsel", "mircuts",
                 "parallelmode", "NumSteps", "nr_regions", "NumCars", "nr_obstacles", "max_lines_obstacles",
                 "nr_environments"]
    VEC_N = ["agent_safety_distance", "agent_safety_distance_slack"]
    VEC_C = ["WEIGHTS_POS_X", "WEIGHTS_VEL_X", "WEIGHTS_ACC_X", "WEIGHTS_POS_Y", "WEIGHTS_VEL_Y", "WEIGHTS_ACC_Y",
             "WEIGHTS_JERK_X", "WEIGHTS_JERK_Y", "WheelBase", "CollisionRadius"]
    MAT_CN = ["x_ref", "vx_ref", "y_ref", "vy_ref"]
    MAT_CR = ["min_acc_x", "max_acc_x", "min_acc_y", "max_acc_y", "min_jerk_x", "max_jerk_x", "min_jerk_y",
              "max_jerk_y"]
    MAT_R3 = ["POLY_SINT_UB", "POLY_SINT_LB", "POLY_COSS_UB", "POLY_COSS_LB", "POLY_KAPPA_AX_MAX",
              "POLY_KAPPA_AX_MIN"]

    def __init__(self):
        for n in self.SCALARS_F:
            setattr(self, n, 0.0)
        for n in self.SCALARS_I:
            setattr(self, n, 0)
        self.max_solution_time = 10.0
        self.relative_mip_gap_tolerance = 0.1
        self.mipdisplay = 2
        self.MultiEnvironmentConvexPolygon = []
        self.ObstacleConvexPolygon = []
        self.obstacle_is_soft = []

    def copy(self):
        import copy
        return copy.deepcopy(self)

    def to_c(self):
        """returns (struct, keepalive list)"""
        s = ModelParamsC()
        keep = []
        for n in self.SCALARS_F:
            setattr(s, n, float(getattr(self, n)))
        for n in self.SCALARS_I:
            setattr(s, n, int(getattr(self, n)))
        Cn, N, R = self.NumCars, self.NumSteps, self.nr_regions

        def dv(name, shape):
            a = np.ascontiguousarray(np.asarray(getattr(self, name), dtype=np.float64).reshape(shape))
            keep.append(a)
            setattr(s, name, _dp(a))
        for n in self.VEC_N:
            dv(n, (N,))
        for n in self.VEC_C:
            dv(n, (Cn,))
        dv("IntitialState", (Cn, 6))
        for n in self.MAT_CN:
            dv(n, (Cn, N))
        for n in self.MAT_CR:
            dv(n, (Cn, R))
        dv("fraction_parameters", (R, 4))
        for n in self.MAT_R3:
            dv(n, (R, 3))
        ir = np.ascontiguousarray(np.asarray(self.initial_region, dtype=np.int32).reshape(Cn))
        pr = np.ascontiguousarray(np.asarray(self.possible_region, dtype=np.int32).reshape(Cn, R))
        keep += [ir, pr]
        s.initial_region = _ip(ir)
        s.possible_region = _ip(pr)
        O, L, E = self.nr_obstacles, self.max_lines_obstacles, self.nr_environments
        ov = np.zeros((max(O, 0), N, max(L, 0), 2))
        for o in range(O):
            for i in range(N):
                ov[o, i] = np.asarray(self.ObstacleConvexPolygon[o][i], dtype=np.float64).reshape(L, 2)
        ov = np.ascontiguousarray(ov)
        so = np.ascontiguousarray(np.asarray(list(self.obstacle_is_soft) + [0], dtype=np.int32))
        offs = [0]
        verts = []
        for e in range(E):
            v = np.asarray(self.MultiEnvironmentConvexPolygon[e], dtype=np.float64).reshape(-1, 2)
            verts.append(v)
            offs.append(offs[-1] + len(v))
        eo = np.ascontiguousarray(np.asarray(offs, dtype=np.int32))
        ev = np.ascontiguousarray(np.concatenate(verts, 0) if verts else np.zeros((1, 2)))
        keep += [ov, so, eo, ev]
        s.obstacle_vertices = _dp(ov)
        s.obstacle_is_soft = _ip(so)
        s.env_offsets = _ip(eo)
        s.env_vertices = _dp(ev)
        return s, keep

    @staticmethod
    def from_dat_dict(d):
        """ModelParameters from a parsed OPL .dat (edge tuples -> vertex lists: x1,y1 of every edge)."""
        p = ModelParameters()
        for n in ModelParameters.SCALARS_F:
            setattr(p, n, float(d[n]))
        for n in ModelParameters.SCALARS_I:
            setattr(p, n, int(d[n]))
        Cn, N, R = p.NumCars, p.NumSteps, p.nr_regions
        for n in ModelParameters.VEC_N + ModelParameters.VEC_C:
            setattr(p, n, np.asarray(d[n], float).reshape(-1))
        p.IntitialState = np.asarray(d["IntitialState"], float).reshape(Cn, 6)
        for n in ModelParameters.MAT_CN:
            setattr(p, n, np.asarray(d[n], float).reshape(Cn, N))

        def widen(a, rows, width):
            a = np.asarray(a, float)
            a = a.reshape(rows, -1) if a.size else np.zeros((rows, 0))
            out = np.zeros((rows, width))
            out[:, :a.shape[1]] = a[:, :width]
            return out
        for n in ModelParameters.MAT_CR:
            setattr(p, n, widen(d[n], Cn, R))
        p.initial_region = np.asarray(d["initial_region"], int).reshape(Cn)
        p.possible_region = widen(d["possible_region"], Cn, R).astype(int)

        def regtab(name, w):
            a = np.asarray(d[name], float).reshape(-1, w)
            out = np.zeros((R, w))
            out[:len(a)] = a[:R]
            return out
        p.fraction_parameters = regtab("fraction_parameters", 4)
        for n in ModelParameters.MAT_R3:
            setattr(p, n, regtab(n, 3))
        p.MultiEnvironmentConvexPolygon = [np.array([[t[1], t[2]] for t in e], float)
                                           for e in d["MultiEnvironmentConvexPolygon"]]
        p.ObstacleConvexPolygon = [[np.array([[t[1], t[2]] for t in poly], float) for poly in o]
                                   for o in d["ObstacleConvexPolygon"]]
        p.obstacle_is_soft = [int(v) for v in d["obstacle_is_soft"]]
        return p


class RawResults:
    """numpy-backed RawResults (src/miqp_planner_data.hpp:46-97), row-major [C,N], [C,E,N], [C,N,R], ..."""

    def __init__(self, Cn, N, R, E, O, L):
        K = Cn - 1
        self.dims = (Cn, N, R, E, O, L)
        self.N, self.NrEnvironments, self.NrRegions, self.NrObstacles = N, E, R, O
        self.MaxLinesObstacles, self.NrCarToCarCollisions, self.NrCars = L, K, Cn
        for n in _RES_D:
            setattr(self, n, np.full((Cn, N), 9999999.0))
        shapes = {"active_region": (Cn, N, R), "deltacc": (Cn, O, N, L), "deltacc_front": (Cn, O, N, L, 4),
                  "car2car_collision": (K, K, N, 16), "slackvars": (K, K, N, 4), "slackvarsObstacle": (Cn, O, N),
                  "slackvarsObstacle_front": (Cn, O, N, 4)}
        for n in _RES_I:
            if n.startswith("notWithin"):
                sh = (Cn, E, N)
            elif n.startswith("region_change"):
                sh = (Cn, N)
            else:
                sh = shapes[n]
            setattr(self, n, np.full(sh, 9999999, dtype=np.int32))
        self.slackvars_real = np.zeros((K, K, N, 4))

    def to_c(self):
        s = RawResultsC()
        Cn, N, R, E, O, L = self.dims
        s.N, s.NrEnvironments, s.NrRegions, s.NrObstacles, s.MaxLinesObstacles = N, E, R, O, L
        s.NrCarToCarCollisions, s.NrCars = Cn - 1, Cn
        for n in _RES_D:
            a = getattr(self, n)
            assert a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
            setattr(s, n, _dp(a))
        for n in _RES_I:
            a = getattr(self, n)
            assert a.flags["C_CONTIGUOUS"] and a.dtype == np.int32
            setattr(s, n, _ip(a))
        s.slackvars_real = _dp(self.slackvars_real)
        return s

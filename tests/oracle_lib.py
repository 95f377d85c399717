"""ctypes binding of the CPU oracle (oracle/oracle.h) for the tests."""
import ctypes as C

import numpy as np

from planner_miqp_amd.ctypes_types import ModelParamsC, RawResults, RawResultsC, SolutionPropertiesC


class OrcSizes(C.Structure):
    _fields_ = [("rows", C.c_int), ("bin", C.c_int), ("cont", C.c_int), ("nnz", C.c_int)]


class OrcOpts(C.Structure):
    _fields_ = [("gap", C.c_double), ("time_limit", C.c_double), ("max_nodes", C.c_longlong), ("verbose", C.c_int)]


class Oracle:
    def __init__(self, path):
        L = C.CDLL(path)
        vp = C.c_void_p
        L.orc_from_params.restype = vp; L.orc_from_params.argtypes = [C.POINTER(ModelParamsC), C.c_int]
        L.orc_from_dat.restype = vp; L.orc_from_dat.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_free.restype = None; L.orc_free.argtypes = [vp]
        L.orc_raw_sizes.restype = OrcSizes; L.orc_raw_sizes.argtypes = [vp]
        L.orc_raw_eval.restype = C.c_double
        L.orc_raw_eval.argtypes = [vp, C.POINTER(RawResultsC), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_char_p, C.c_int]
        L.orc_solve.restype = C.c_int
        L.orc_solve.argtypes = [vp, C.POINTER(OrcOpts), C.POINTER(RawResultsC), C.POINTER(SolutionPropertiesC)]
        L.orc_solve_fixed.restype = C.c_int
        L.orc_solve_fixed.argtypes = [vp, C.POINTER(RawResultsC), C.POINTER(RawResultsC), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        self.L = L

    # ---- instances
    def from_dat(self, path):
        err = C.create_string_buffer(256)
        h = self.L.orc_from_dat(path.encode(), err, 256)
        if not h:
            raise RuntimeError(err.value.decode())
        return h

    def from_params(self, params, round_decimals=10):
        s, keep = params.to_c()
        h = self.L.orc_from_params(C.byref(s), round_decimals)
        if not h:
            raise RuntimeError("orc_from_params failed")
        return h

    def free(self, h):
        self.L.orc_free(h)

    def dims(self, params):
        return (params.NumCars, params.NumSteps, params.nr_regions, params.nr_environments, params.nr_obstacles,
                params.max_lines_obstacles)

    def sizes(self, h):
        s = self.L.orc_raw_sizes(h)
        return dict(rows=s.rows, bin=s.bin, cont=s.cont, nnz=s.nnz)

    def raw_eval(self, h, res: RawResults, use_real_slack=True):
        rc = res.to_c()
        obj = C.c_double(0)
        w = C.create_string_buffer(96)
        sl = res.slackvars_real.ctypes.data_as(C.POINTER(C.c_double)) if use_real_slack else None
        v = self.L.orc_raw_eval(h, C.byref(rc), sl, C.byref(obj), w, 96)
        return v, obj.value, w.value.decode()

    def solve(self, h, dims, gap=-1.0, time_limit=0.0, max_nodes=0, verbose=0):
        res = RawResults(*dims)
        rc = res.to_c()
        p = SolutionPropertiesC()
        o = OrcOpts(gap, time_limit, max_nodes, verbose)
        st = self.L.orc_solve(h, C.byref(o), C.byref(rc), C.byref(p))
        return st, res, p

    def solve_fixed(self, h, dims, fixed: RawResults):
        res = RawResults(*dims)
        rc = res.to_c()
        fc = fixed.to_c()
        obj = C.c_double(0)
        it = C.c_int(0)
        st = self.L.orc_solve_fixed(h, C.byref(fc), C.byref(rc), C.byref(obj), C.byref(it))
        return st, res, obj.value, it.value

"""Python mirror of miqp::planner::cplex::CplexWrapper (src/cplex_wrapper.hpp:61-275) over libmiqp_gpu.so.

Method names, argument meaning and error behaviour follow the reference class so that parity tests read
like test/cplex_wrapper_test.cc.  All solving happens in the HIP library; this file only marshals."""
import ctypes as C
import enum
import os
import subprocess

import numpy as np

# (four concurrent launches per round + the null stream: see miqp_gpu.hip - effective when set before the process's first HIP call)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .ctypes_types import (ModelParameters, ModelParamsC, RawResults, RawResultsC, SolutionPropertiesC, SolverOptsC, c_double_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# miqp_exchange_fn (include/miqp_gpu.h): int (*)(void* user, int op, void* buf, int count, int root)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int)


def library_path():
    return os.environ.get("MIQP_GPU_LIB", os.path.join(_HERE, "libmiqp_gpu.so"))


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of the in-tree shared library (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc", "miqp_gpu.hip")
    out = library_path()
    deps = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    deps += [os.path.join(_HERE, "..", "include", f) for f in ("miqp_gpu.h", "miqp_types.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the value-changing fast-math switches apply to the DEVICE code only: the host side of the translation unit holds the
    # restatement of RoundWithPrecision (host_inst.hpp::round_dec), which has to divide exactly as the reference does
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fno-math-errno", "-fno-trapping-math",
           "-Xarch_device", "-freciprocal-math", "-Xarch_device", "-fno-signed-zeros",
           "-fPIC", "-shared", "-std=c++17", "-o", out, src]
    subprocess.check_call(cmd)
    return out


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError("libmiqp_gpu.so is not built (run __graft_entry__.build()); the solver has no CPU fallback")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.miqp_solver_create.restype = vp; L.miqp_solver_create.argtypes = [C.POINTER(SolverOptsC)]
    L.miqp_solver_destroy.restype = None; L.miqp_solver_destroy.argtypes = [vp]
    L.miqp_solver_set_params.restype = C.c_int; L.miqp_solver_set_params.argtypes = [vp, C.POINTER(ModelParamsC)]
    L.miqp_solver_load_dat.restype = C.c_int; L.miqp_solver_load_dat.argtypes = [vp, C.c_char_p]
    L.miqp_solver_override_settings.restype = C.c_int; L.miqp_solver_override_settings.argtypes = [vp, C.c_double, C.c_double]
    L.miqp_solver_set_warmstart.restype = C.c_int; L.miqp_solver_set_warmstart.argtypes = [vp, C.POINTER(RawResultsC), C.c_int]
    L.miqp_solver_solve.restype = C.c_int; L.miqp_solver_solve.argtypes = [vp, C.c_double]
    L.miqp_solver_solve_batch.restype = C.c_int; L.miqp_solver_solve_batch.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    L.miqp_solver_solve_stream.restype = C.c_int; L.miqp_solver_solve_stream.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.miqp_solver_solve_batch_multi.restype = C.c_int; L.miqp_solver_solve_batch_multi.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.miqp_solver_raw_sizes.restype = C.c_int; L.miqp_solver_raw_sizes.argtypes = [vp, C.POINTER(C.c_int)]
    L.miqp_solver_lift_tables.restype = C.c_int; L.miqp_solver_lift_tables.argtypes = [vp, c_double_p, C.c_int]
    L.miqp_solver_solve_split.restype = C.c_int; L.miqp_solver_solve_split.argtypes = [vp, C.c_double, C.c_int, C.c_int, EXCHANGE_FN, vp]
    L.miqp_solver_solve_split_rccl.restype = C.c_int; L.miqp_solver_solve_split_rccl.argtypes = [vp, C.c_double]
    L.miqp_solver_split_roots.restype = C.c_int
    L.miqp_solver_split_roots.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.miqp_comm_unique_id.restype = C.c_int; L.miqp_comm_unique_id.argtypes = [C.c_char_p]
    L.miqp_comm_init.restype = C.c_int; L.miqp_comm_init.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.miqp_comm_finalize.restype = C.c_int; L.miqp_comm_finalize.argtypes = []
    L.miqp_comm_selftest.restype = C.c_int; L.miqp_comm_selftest.argtypes = [EXCHANGE_FN, vp, C.c_int, C.c_int]
    L.miqp_solver_get_results.restype = C.c_int; L.miqp_solver_get_results.argtypes = [vp, C.POINTER(RawResultsC)]
    L.miqp_solver_materialize_results.restype = C.c_int; L.miqp_solver_materialize_results.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.miqp_solver_get_properties.restype = C.c_int; L.miqp_solver_get_properties.argtypes = [vp, C.POINTER(SolutionPropertiesC)]
    L.miqp_solver_get_dims.restype = C.c_int; L.miqp_solver_get_dims.argtypes = [vp, C.POINTER(C.c_int)]
    L.miqp_solver_export_lp.restype = C.c_int; L.miqp_solver_export_lp.argtypes = [vp, C.c_char_p]
    for fn in (L.miqp_solver_write_dat, L.miqp_solver_write_solution, L.miqp_solver_write_mst, L.miqp_solver_read_mst):
        fn.restype = C.c_int; fn.argtypes = [vp, C.c_char_p]
    L.miqp_solver_solve_fixed.restype = C.c_int
    L.miqp_solver_solve_fixed.argtypes = [vp, C.POINTER(RawResultsC), C.POINTER(RawResultsC), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.miqp_solver_last_timing.restype = C.c_int; L.miqp_solver_last_timing.argtypes = [vp, C.POINTER(C.c_double)]
    L.miqp_solver_last_setup.restype = C.c_int; L.miqp_solver_last_setup.argtypes = [vp, C.POINTER(C.c_double)]
    L.miqp_solver_last_active_set.restype = C.c_int; L.miqp_solver_last_active_set.argtypes = [vp, C.POINTER(C.c_double)]
    L.miqp_solver_last_error.restype = C.c_char_p; L.miqp_solver_last_error.argtypes = [vp]
    L.miqp_solver_last_admission.restype = C.c_int; L.miqp_solver_last_admission.argtypes = [vp, C.POINTER(C.c_double)]
    L.miqp_gpu_version.restype = C.c_char_p
    _LIB = L
    return L


EXPORTED_SYMBOLS = ["miqp_solver_create", "miqp_solver_destroy", "miqp_solver_set_params", "miqp_solver_load_dat",
                    "miqp_solver_override_settings", "miqp_solver_set_warmstart", "miqp_solver_solve",
                    "miqp_solver_solve_batch", "miqp_solver_get_results", "miqp_solver_get_properties",
                    "miqp_solver_get_dims", "miqp_solver_export_lp", "miqp_solver_solve_fixed",
                    "miqp_solver_last_timing", "miqp_solver_last_active_set", "miqp_solver_last_setup", "miqp_solver_last_error", "miqp_solver_last_admission", "miqp_gpu_version", "miqp_solver_write_dat", "miqp_solver_write_solution",
                    "miqp_solver_write_mst", "miqp_solver_read_mst", "miqp_fraction_parameters", "miqp_mean_angles",
                    "miqp_limits_per_region", "miqp_calculate_region_idx", "miqp_reserve_neighbor_regions",
                    "miqp_calculate_possible_regions", "miqp_calculate_warmstart", "miqp_plan",
                    "miqp_solver_solve_batch_multi", "miqp_solver_raw_sizes", "miqp_solver_lift_tables", "miqp_reference_trajectory", "miqp_update_car", "miqp_fitting_polynomial_parameters",
                    "miqp_solver_solve_split", "miqp_solver_solve_split_rccl", "miqp_solver_split_roots", "miqp_comm_unique_id",
                    "miqp_comm_init", "miqp_comm_finalize", "miqp_comm_selftest", "miqp_solver_solve_stream", "miqp_solver_materialize_results",
                    "miqp_initial_pose_check", "miqp_select_environment", "miqp_obstacle_intersects_environment", "miqp_obstacles_roi", "miqp_bark_trajectory", "miqp_obstacle_intersects_environment_roi", "miqp_environment_warmstart"]


class OptimizationStatus(enum.IntEnum):  # src/cplex_wrapper.hpp:54-59
    SUCCESS = 0
    FAILED_NO_SOLUT = 1
    FAILED_SEG_FAULT = 2
    FAILED_TIMEOUT = 3


class WarmstartType(enum.IntEnum):  # src/miqp_planner_settings.h:13-18
    NO_WARMSTART = 0
    RECEDING_HORIZON_WARMSTART = 1
    LAST_SOLUTION_WARMSTART = 2
    BOTH_WARMSTART_STRATEGIES = 3


class ParameterSource(enum.IntEnum):  # src/cplex_wrapper.hpp:63
    DATFILE = 0
    CPPINPUTS = 1
    MIXED = 2


class SolutionProperties:  # src/cplex_wrapper.hpp:41-52
    def __init__(self, c=None):
        for n, _ in SolutionPropertiesC._fields_:
            setattr(self, n, getattr(c, n) if c is not None else 0)

    def __repr__(self):
        return "SolutionProperties(" + ", ".join("%s=%r" % (n, getattr(self, n)) for n, _ in SolutionPropertiesC._fields_) + ")"


class CplexWrapper:
    """Same public surface as the reference class; ``modfile`` is accepted and ignored (the OPL model is
    built into the device solver)."""

    def __init__(self, modfile="cplexmodel.mod", parameterSource=ParameterSource.CPPINPUTS, precision=12, modpath="cplexmodel/",
                 nodes_per_round=0, max_open_nodes=0, gap_override=-1.0, verbose=0, device=-1):
        self._L = load_library()
        self._opts = SolverOptsC(int(precision), int(device), int(nodes_per_round), int(max_open_nodes), float(gap_override), int(verbose))
        self._h = self._L.miqp_solver_create(C.byref(self._opts))
        self.parameterSource_ = ParameterSource(parameterSource)
        self.modfile_ = modpath + modfile
        self.datfile_ = ""
        self._params = None
        self._keep = None
        self._results = None
        self._warm = None
        self.useSpecialOrderedSets_ = False
        self.useBranchingPriorities_ = False
        self.doWarmstart_ = WarmstartType.NO_WARMSTART
        self.debugOutputFilePath_ = ""
        self.debugOutputFilePrefix_ = ""
        self.print_debug_outputs_ = False
        self.debugOutputParameterFilePath_ = ""
        self.tmpWarmstartFile_ = "/tmp/warmstart_debug_res.mst"   # src/cplex_wrapper.hpp:104 (shared by all instances)

    def __del__(self):
        try:
            if self._h:
                self._L.miqp_solver_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- parameters
    def setParameterDatFileRelative(self, datfile):
        self.datfile_ = "cplexmodel/" + datfile

    def setParameterDatFileAbsolute(self, datfile):
        self.datfile_ = datfile

    def resetParameters(self, parameters: ModelParameters):
        self._params = parameters  # shared with the caller, re-read on every callCplex (cplex_wrapper.cpp:661-664)

    def overrideSolverSettingsDataSource(self, parameters: ModelParameters):
        if self._params is not None:
            for n in ["max_solution_time", "relative_mip_gap_tolerance", "mipdisplay", "mipemphasis", "relobjdif", "cutpass",
                      "probe", "repairtries", "rinsheur", "varsel", "mircuts", "parallelmode"]:
                setattr(self._params, n, getattr(parameters, n))

    # ---- options that only steer CPLEX's search (accepted, no effect on the result: K7)
    def setSpecialOrderedSets(self, v):
        self.useSpecialOrderedSets_ = bool(v)

    def setUseBranchingPriorities(self, v):
        self.useBranchingPriorities_ = bool(v)

    def setBranchingPriorityValueExtent(self, value, extent):
        self.branchingPriority_ = (value, extent)

    def setBufferCplexOutputsToStream(self, v):
        pass

    def setDebugOutputPrint(self, v):
        self.print_debug_outputs_ = bool(v)

    def setDebugOutputFilePath(self, p):
        self.debugOutputFilePath_ = p

    def setDebugOutputFilePrefix(self, p):
        self.debugOutputFilePrefix_ = p

    # ---- warm start
    def addRecedingHorizonWarmstart(self, warmstart: RawResults, wt=WarmstartType.RECEDING_HORIZON_WARMSTART):
        self.doWarmstart_ = wt if wt == WarmstartType.RECEDING_HORIZON_WARMSTART else WarmstartType.BOTH_WARMSTART_STRATEGIES
        self._warm = warmstart

    def setLastSolutionWarmstart(self, wt=WarmstartType.LAST_SOLUTION_WARMSTART):
        self.doWarmstart_ = wt if wt == WarmstartType.LAST_SOLUTION_WARMSTART else WarmstartType.BOTH_WARMSTART_STRATEGIES

    def deleteLastSolutionWarmstartFile(self):
        """src/cplex_wrapper.cpp:482-488"""
        self._last = None
        if os.path.exists(self.tmpWarmstartFile_):
            os.remove(self.tmpWarmstartFile_)

    # ---- solve
    def _push_inputs(self):
        if self.parameterSource_ == ParameterSource.DATFILE:
            rc = self._L.miqp_solver_load_dat(self._h, self.datfile_.encode())
        elif self.parameterSource_ == ParameterSource.MIXED and self._params is None:
            rc = self._L.miqp_solver_load_dat(self._h, self.datfile_.encode())   # MIXED: the C++ inputs when given, else the file
        else:
            if self._params is None:
                return -1
            s, self._keep = self._params.to_c()
            rc = self._L.miqp_solver_set_params(self._h, C.byref(s))
        if rc != 0:
            return rc
        # MIP starts (src/cplex_wrapper.cpp:121-138): the receding-horizon start and, independently of it, the .mst
        # file of the last solution - with BOTH_WARMSTART_STRATEGIES the reference applies both
        self._L.miqp_solver_set_warmstart(self._h, None, int(WarmstartType.NO_WARMSTART))
        if self.doWarmstart_ in (WarmstartType.RECEDING_HORIZON_WARMSTART, WarmstartType.BOTH_WARMSTART_STRATEGIES) and self._warm is not None:
            wc = self._warm.to_c()
            self._L.miqp_solver_set_warmstart(self._h, C.byref(wc), int(WarmstartType.RECEDING_HORIZON_WARMSTART))
        if self.doWarmstart_ in (WarmstartType.LAST_SOLUTION_WARMSTART, WarmstartType.BOTH_WARMSTART_STRATEGIES):
            # cplex.readMIPStarts(tmpWarmstartFile_) when the file exists (src/cplex_wrapper.cpp:128-138)
            if os.path.exists(self.tmpWarmstartFile_):
                self._L.miqp_solver_read_mst(self._h, self.tmpWarmstartFile_.encode())
        return 0

    def _stamp(self, timestamp):
        return "%.15g" % float(timestamp)

    def _debug_before(self, timestamp):
        """parameters_<t>.txt (OPL external data) and lpexport_<t>.lp (src/cplex_wrapper.cpp:141-155)"""
        if not self.print_debug_outputs_:
            return
        base = os.path.join(self.debugOutputFilePath_, self.debugOutputFilePrefix_)
        self.debugOutputParameterFilePath_ = base + "parameters_" + self._stamp(timestamp) + ".txt"
        self._L.miqp_solver_write_dat(self._h, self.debugOutputParameterFilePath_.encode())
        self._L.miqp_solver_export_lp(self._h, (base + "lpexport_" + self._stamp(timestamp) + ".lp").encode())

    def _debug_after(self, timestamp, status):
        """MIP start of the last solution and solution_<t>.txt / warmstartsolution_<t>.mst (src/cplex_wrapper.cpp:206-229)"""
        if status != OptimizationStatus.SUCCESS:
            return
        last = self.doWarmstart_ in (WarmstartType.LAST_SOLUTION_WARMSTART, WarmstartType.BOTH_WARMSTART_STRATEGIES)
        if last:
            self._L.miqp_solver_write_mst(self._h, self.tmpWarmstartFile_.encode())
        if self.print_debug_outputs_:
            base = os.path.join(self.debugOutputFilePath_, self.debugOutputFilePrefix_)
            self._L.miqp_solver_write_solution(self._h, (base + "solution_" + self._stamp(timestamp) + ".txt").encode())
            if last:
                self._L.miqp_solver_write_mst(self._h, (base + "warmstartsolution_" + self._stamp(timestamp) + ".mst").encode())

    def _collect(self, status, lazy=False):
        """takes the result record over from the library; lazy (batch entry points): on the first getRawResults()"""
        self._stale = status == OptimizationStatus.SUCCESS
        if not self._stale:
            return OptimizationStatus(status)
        if not lazy:
            self._fetch()
        return OptimizationStatus(status)

    def _fetch(self):
        if getattr(self, "_stale", False):
            d = (C.c_int * 6)()
            self._L.miqp_solver_get_dims(self._h, d)
            res = RawResults(*list(d))
            rc = res.to_c()
            if self._L.miqp_solver_get_results(self._h, C.byref(rc)) != 0:
                raise RuntimeError("miqp_solver_get_results failed: the library holds no solution for this handle")
            self._results = res
            self._last = res
            self._stale = False

    def callCplex(self, timestamp=0.0):
        if self._push_inputs() != 0:
            return OptimizationStatus.FAILED_SEG_FAULT
        self._debug_before(timestamp)
        st = self._collect(self._L.miqp_solver_solve(self._h, float(timestamp)))
        self._debug_after(timestamp, st)
        return st

    def callCplexSplit(self, world, rank, exchange=None, timestamp=0.0):
        """one instance whose branch-and-bound tree is split over the ranks of a job (miqp_solver_solve_split): every rank
        calls this with the same parameters; ``exchange`` is an EXCHANGE_FN (e.g. sharding.torch_exchange()), None = the RCCL
        communicator of sharding.init_rccl_comm().  Every rank returns the same status and holds the full result."""
        if self._push_inputs() != 0:
            return OptimizationStatus.FAILED_SEG_FAULT
        if exchange is None:
            st = self._L.miqp_solver_solve_split_rccl(self._h, float(timestamp))
        else:
            st = self._L.miqp_solver_solve_split(self._h, float(timestamp), int(world), int(rank), exchange, None)
        return self._collect(st)

    def splitRoots(self, world, rank):
        """the roots of rank ``rank`` in a tree split over ``world`` ranks: (list of roots, each a list of (record index,
        alternative) fixings; size of the partition).  No device needed."""
        if self._push_inputs() != 0:
            return None
        cap = 4096
        ro, ix, va = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
        nr, nc = C.c_int(0), C.c_int(0)
        n = self._L.miqp_solver_split_roots(self._h, int(world), int(rank), ro, ix, va, cap, C.byref(nr), C.byref(nc))
        if n < 0:
            return None
        roots = [[] for _ in range(nr.value)]
        for k in range(min(n, cap)):
            roots[ro[k]].append((ix[k], va[k]))
        return roots, nc.value

    def getRawResults(self):
        self._fetch()
        return self._results

    def getSolutionProperties(self):
        p = SolutionPropertiesC()
        self._L.miqp_solver_get_properties(self._h, C.byref(p))
        return SolutionProperties(p)

    def getTmpWarmstartFile(self):
        return self.tmpWarmstartFile_

    def getDebugOutputParameterFilePath(self):
        return self.debugOutputParameterFilePath_

    def writeDat(self, path):
        """the parameters of the next solve as OPL external data (no device needed)"""
        if self._push_inputs() != 0:
            return -1
        return self._L.miqp_solver_write_dat(self._h, path.encode())

    # ---- extras of this implementation
    def solveFixed(self, fixed: RawResults):
        """continuous QP with the binaries of ``fixed`` asserted (device interior point kernel)"""
        if self._push_inputs() != 0:
            return None
        d = (C.c_int * 6)()
        self._L.miqp_solver_get_dims(self._h, d)
        out = RawResults(*list(d))
        oc = out.to_c()
        fc = fixed.to_c()
        obj = C.c_double(0)
        it = C.c_int(0)
        rc = self._L.miqp_solver_solve_fixed(self._h, C.byref(fc), C.byref(oc), C.byref(obj), C.byref(it))
        return rc, out, obj.value, it.value

    def liftTables(self):
        """response tables of the bound lifting, array [car][axis][step][4][4] (diagnostic, no device needed)"""
        if self._push_inputs() != 0:
            raise RuntimeError("invalid parameters")
        p = self._params
        out = np.zeros((p.NumCars, 2, p.NumSteps, 4, 4))
        n = self._L.miqp_solver_lift_tables(self._h, out.ctypes.data_as(c_double_p), out.size)
        if n != out.size:
            raise RuntimeError("miqp_solver_lift_tables failed (%d)" % n)
        return out

    def rawSizes(self):
        """rows / binaries / continuous columns / non-zeros of the OPL model of the loaded parameters (no device needed)"""
        if self._push_inputs() != 0:
            return None
        o = (C.c_int * 4)()
        if self._L.miqp_solver_raw_sizes(self._h, o) != 0:
            return None
        return dict(rows=o[0], bin=o[1], cont=o[2], nnz=o[3])

    def lastAdmission(self):
        """seconds after the start of the last batch / stream call at which this instance was admitted to a slot"""
        o = (C.c_double * 1)()
        self._L.miqp_solver_last_admission(self._h, o)
        return o[0]

    def lastError(self):
        """why the last solve of this wrapper did not run or did not finish ("" when there is nothing to say)"""
        return (self._L.miqp_solver_last_error(self._h) or b"").decode()

    def lastTiming(self):
        t = (C.c_double * 6)()
        self._L.miqp_solver_last_timing(self._h, t)
        u = (C.c_double * 3)()
        self._L.miqp_solver_last_setup(self._h, u)
        v = (C.c_double * 8)()
        self._L.miqp_solver_last_active_set(self._h, v)
        return dict(solve_s=t[0], ipm_s=t[1], ipm_launches=int(t[2]), nodes=int(t[3]), ipm_iters=int(t[4]), row_iters=int(t[5]),
                    setup_s=u[0], context_s=u[1], context_built=bool(u[2]),
                    as_nodes=int(v[0]), as_steps=int(v[1]), as_unfinished=int(v[2]), as_drops=int(v[3]), as_rows_end=int(v[4]), as_rows_parent=int(v[5]),
                    std_launch_s=v[6], std_launches=int(v[7]))


def prepare_batch(wrappers):
    """hands the parameters (and MIP starts) of every wrapper to its solver handle - the marshalling part of a batch call, which
    a caller with many instances does while it builds them (bench.py: before the timed region)"""
    for w in wrappers:
        if w._push_inputs() != 0:
            raise RuntimeError("invalid parameters")


def solve_batch(wrappers, gpus=None, inflight=None, prepared=False):
    """Solves independent instances concurrently: on one device (miqp_solver_solve_batch), or with ``gpus`` given
    sharded b -> device b mod gpus inside the library (miqp_solver_solve_batch_multi; 0 = every visible device).
    ``inflight``: the call is a queue drained with that many instances in flight (miqp_solver_solve_stream);
    ``prepared``: prepare_batch(wrappers) has been called already.  Result records are fetched on the first getRawResults()."""
    L = load_library()
    if not prepared:
        prepare_batch(wrappers)
    n = len(wrappers)
    hs = (C.c_void_p * n)(*[w._h for w in wrappers])
    st = (C.c_int * n)(*([int(OptimizationStatus.FAILED_SEG_FAULT)] * n))   # (a status the library does not write must not read as SUCCESS = 0)
    if inflight is not None and gpus is None:
        rc = L.miqp_solver_solve_stream(hs, n, int(inflight), st)
    else:
        rc = L.miqp_solver_solve_batch(hs, n, st) if gpus is None else L.miqp_solver_solve_batch_multi(hs, n, int(gpus), st)
    if rc not in (0, -2):
        return [OptimizationStatus.FAILED_SEG_FAULT] * n
    # (-2: the call failed as a whole or in part - the library has set every status: FAILED_SEG_FAULT for the instances it did not run)
    return [w._collect(st[k], lazy=True) for k, w in enumerate(wrappers)]


def materialize_results(wrappers, threads=0):
    """builds the RawResults records of a solved batch on host threads inside the library (miqp_solver_materialize_results);
    getRawResults() of each wrapper then only copies.  Returns the number of records built."""
    L = load_library()
    n = len(wrappers)
    hs = (C.c_void_p * n)(*[w._h for w in wrappers])
    return L.miqp_solver_materialize_results(hs, n, int(threads))

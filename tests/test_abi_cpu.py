"""CPU-side checks: the C-ABI library loads and exports every symbol of include/miqp_gpu.h, host logic,
sharding over gloo.  No compute call needs a GPU here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import planner_miqp_amd as P
from helpers import dat_path, load_params
from planner_miqp_amd import synthetic
from planner_miqp_amd.sharding import shard_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    P.build_library()
    return P.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "miqp_gpu.h")).read()
    names = set(re.findall(r"\b(miqp_[a-z_]+)\s*\(", hdr))
    assert names, "no prototypes found"
    for n in names:
        assert hasattr(lib, n), n
    from planner_miqp_amd.wrapper import EXPORTED_SYMBOLS
    assert names == set(EXPORTED_SYMBOLS)


def test_set_params_and_dims(lib):
    import ctypes as C
    p = load_params("cplexmodel_testcase.dat")
    w = P.CplexWrapper()
    w.resetParameters(p)
    assert w._push_inputs() == 0
    d = (C.c_int * 6)()
    assert lib.miqp_solver_get_dims(w._h, d) == 0
    assert list(d) == [1, 20, 32, 1, 1, 4]


def test_load_dat_errors(lib):
    w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE)
    w.setParameterDatFileAbsolute("/nonexistent/file.dat")
    assert w.callCplex() == P.OptimizationStatus.FAILED_SEG_FAULT


@pytest.mark.parametrize("name", ["cplexmodel_testcase.dat", "cplexmodel.dat", "test_sos.dat"])
def test_dat_writer_round_trip(lib, tmp_path, name):
    """printExternalData equivalent (src/cplex_wrapper.cpp:141-149): fixture -> written .dat -> re-read gives the same
    instance; compared through the LP dump of the raw model, which depends on every parameter"""
    w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE)
    w.setParameterDatFileAbsolute(dat_path(name))
    out = str(tmp_path / "written.dat")
    assert w.writeDat(out) == 0
    lp1, lp2 = str(tmp_path / "a.lp"), str(tmp_path / "b.lp")
    assert lib.miqp_solver_export_lp(w._h, lp1.encode()) == 0
    w2 = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE)
    w2.setParameterDatFileAbsolute(out)
    assert w2._push_inputs() == 0
    assert lib.miqp_solver_export_lp(w2._h, lp2.encode()) == 0
    assert open(lp1).read() == open(lp2).read()
    # the numpy .dat reader of the test tools accepts the written file too
    from miqp_py.dat import load_dat
    d = load_dat(out)
    assert int(d["NumSteps"]) == load_params(name).NumSteps


def test_no_cpu_fallback(lib):
    """without a HIP device the product must fail loudly, never solve on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE)
    w.setParameterDatFileAbsolute(dat_path("cplexmodel_testcase.dat"))
    assert w.callCplex() == P.OptimizationStatus.FAILED_SEG_FAULT
    assert w.getRawResults() is None


def test_product_does_not_reference_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "planner_miqp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in txt.replace("no dependency on oracle/", "") and "liboracle" not in txt and "oracle_lib" not in txt, f


def test_synthetic_generator_is_deterministic_and_valid(oracle):
    a = synthetic.generate("cfg3", 7)
    b = synthetic.generate("cfg3", 7)
    assert np.array_equal(a.IntitialState, b.IntitialState) and np.array_equal(a.x_ref, b.x_ref)
    h = oracle.from_params(a, 10)
    s = oracle.sizes(h)
    assert s["bin"] == 2 * 20 * (5 * 2 + 32 + 5) + 20 * 16 and s["cont"] == 2 * 20 * 12 + 20 * 4
    oracle.free(h)


def test_shard_indices_partition():
    n = 37
    seen = []
    for r in range(4):
        seen += shard_indices(n, r, 4)
    assert sorted(seen) == list(range(n))


def _run_ranks(code, nproc=2, port=29533, timeout=600):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
        script = f.name
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1",
                          "--master-port", str(port), script], capture_output=True, text=True, env=env, timeout=timeout)
    os.unlink(script)
    return out


def test_gloo_two_ranks_shard_solve_and_exchange():
    """the N > 1 paths on two gloo ranks (no GPU here, so every solve ends in the loud no-device failure - the sharding, the
    gather and the incumbent-exchange transport are the real code): (1) solve_sharded: instance b -> rank b mod 2, every
    instance reported exactly once and in order on both ranks; (2) the exchange callback the tree split uses, checked by the
    library's own contract test (all-reduce(min) over unsigned words incl. the top bit, broadcast from every root);
    (3) the roots of the two ranks partition the tree; (4) a split solve reaches the transport-independent failure path on
    both ranks without hanging"""
    code = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import planner_miqp_amd as P
from planner_miqp_amd import synthetic, sharding
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
ws = []
for s in range(7):
    x = P.CplexWrapper(); x.resetParameters(synthetic.generate("mini", s)); ws.append(x)
rec = sharding.solve_sharded(ws)
assert len(rec) == 7 and all(q is not None for q in rec), rec
assert all(q["status"] == int(P.OptimizationStatus.FAILED_SEG_FAULT) for q in rec) or torch.cuda.is_available()
g = sharding.gather_counts([float(len(sharding.shard_indices(7, r, w))), float(sum(sharding.shard_indices(7, r, w)))])
assert sum(x[0] for x in g) == 7 and sum(x[1] for x in g) == 21, g
ex = sharding.torch_exchange()
L = P.load_library()
assert L.miqp_comm_selftest(ex, None, w, r) == 0
x = P.CplexWrapper(); x.resetParameters(synthetic.generate("cfg5s", 0))
mine, ncomb = x.splitRoots(w, r)
allr = [None] * w
dist.all_gather_object(allr, mine)
flat = [tuple(q) for part in allr for q in part]
assert len(flat) == ncomb == len(set(flat)) and ncomb >= 4 * w, (ncomb, len(flat))
keys = sorted({k for q in flat for k, _ in q})
import itertools
alts = {k: sorted({v for q in flat for kk, v in q if kk == k}) for k in keys}
assert sorted(flat) == sorted(tuple(zip(keys, c)) for c in itertools.product(*[alts[k] for k in keys])), "not a partition"
if not torch.cuda.is_available():
    st = sharding.split_solve(x, ex)
    assert st == P.OptimizationStatus.FAILED_SEG_FAULT
if r == 0:
    print("SHARD_OK")
dist.destroy_process_group()
""" % ROOT
    out = _run_ranks(code)
    assert "SHARD_OK" in out.stdout, out.stdout + out.stderr


def test_lp_export_has_the_reference_model_sizes(lib, tmp_path):
    """miqp_solver_export_lp (cplex.exportModel, src/cplex_wrapper.cpp:150-154): the LP dump of the testcase has the
    row and binary counts CPLEX reports for the reference model (test/cplex_wrapper_test.cc:866-871)"""
    w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE)
    w.setParameterDatFileAbsolute(dat_path("cplexmodel_testcase.dat"))
    assert w._push_inputs() == 0
    out = str(tmp_path / "testcase.lp")
    assert lib.miqp_solver_export_lp(w._h, out.encode()) == 0
    txt = open(out).read()
    body = txt.split("Subject To")[1].split("Bounds")[0]
    rows = [l for l in body.splitlines() if l.strip().startswith("c")]
    assert len(rows) == 12361
    nnz = sum(len(re.findall(r"[+-][0-9.eE+-]+ [A-Za-z_]", l)) for l in rows)
    assert nnz == 29834
    bins = [l for l in txt.split("Binaries")[1].split("End")[0].splitlines() if l.strip()]
    assert len(bins) == 1240


def test_raw_sizes_match_the_reference_and_the_oracle_enumeration(lib, oracle):
    """SolutionProperties.NrConstraints / NrBinaryVariables / NrFloatVariables / NonZeroCoefficients
    (collectCplexStatistics, src/cplex_wrapper.cpp:679-690): the product counts them per .mod statement; the reference
    asserts 12361 / 1240 / 340 / 29834 for the testcase (test/cplex_wrapper_test.cc:866-871) and the oracle's row-by-row
    enumeration gives the same four numbers for the other fixtures and for seeded instances of every config"""
    w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE)
    w.setParameterDatFileAbsolute(dat_path("cplexmodel_testcase.dat"))
    assert w.rawSizes() == dict(rows=12361, bin=1240, cont=340, nnz=29834)
    for name in ("cplexmodel.dat", "test_sos.dat"):
        w = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE)
        w.setParameterDatFileAbsolute(dat_path(name))
        h = oracle.from_dat(dat_path(name))
        assert w.rawSizes() == oracle.sizes(h), name
        oracle.free(h)
    for cfg, seed in (("cfg2", 0), ("cfg3", 1), ("cfg4", 2), ("cfg5", 0), ("mini3", 3), ("mini1", 4), ((2, 6, 16, 2, 1), 0), ((1, 12, 32, 0, 0), 1)):
        p = synthetic.generate(cfg, seed)
        if p.nr_obstacles:
            p.obstacle_is_soft = [1] + [0] * (p.nr_obstacles - 1)
        w = P.CplexWrapper(); w.resetParameters(p)
        h = oracle.from_params(p, 10)
        assert w.rawSizes() == oracle.sizes(h), (cfg, seed)
        oracle.free(h)


def test_invalid_records_are_refused(lib):
    """sizes and records that do not fit the instance fail with an error code instead of being indexed"""
    import ctypes as C
    from planner_miqp_amd.ctypes_types import RawResults
    p = synthetic.generate("mini", 0)
    w = P.CplexWrapper(); w.resetParameters(p)
    assert w._push_inputs() == 0
    bad = RawResults(2, 9, 32, 1, 0, 0)                       # another horizon
    assert lib.miqp_solver_set_warmstart(w._h, C.byref(bad.to_c()), 1) < 0
    good = RawResults(2, 8, 32, 1, 0, 0)
    assert lib.miqp_solver_set_warmstart(w._h, C.byref(good.to_c()), 1) == 0
    q = synthetic.generate("mini", 0); q.initial_region = np.array([0, 40])
    w2 = P.CplexWrapper(); w2.resetParameters(q)
    assert w2._push_inputs() != 0
    q = synthetic.generate("mini", 0); q.possible_region = np.zeros_like(q.possible_region)
    w3 = P.CplexWrapper(); w3.resetParameters(q)
    assert w3._push_inputs() != 0
    # ... and the handle says why (miqp_solver_last_error: the LOG(ERROR) of the reference's callCplex as text)
    assert "initial_region" in w2.lastError() and "possible region" in w3.lastError(), (w2.lastError(), w3.lastError())
    assert w.lastError() == ""


def test_a_solve_without_a_device_reports_failed_seg_fault_for_every_instance():
    """no HIP device in this container: every solve entry point fails loudly with the reference's "the solver could not run" code - a status
    array the library never wrote must not read as SUCCESS (= 0), and a handle without a solution refuses to hand out results"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ws = []
    for s in range(3):
        w = P.CplexWrapper(); w.resetParameters(synthetic.generate("mini", s)); ws.append(w)
    for sts in (P.solve_batch(ws), P.solve_batch(ws, inflight=2), P.solve_batch(ws, gpus=1)):
        assert [int(x) for x in sts] == [int(P.OptimizationStatus.FAILED_SEG_FAULT)] * 3
    assert int(ws[0].callCplex()) == int(P.OptimizationStatus.FAILED_SEG_FAULT) and ws[0].getRawResults() is None


def test_all_baseline_configs_generate_and_size(oracle):
    """every BASELINE config of the generator loads into the oracle with the raw sizes of SURVEY App. B"""
    for cfg in ("cfg2", "cfg3", "cfg4", "cfg5", "cfg5s", "mini3b"):
        p = synthetic.generate(cfg, 1)
        Cn, N, R, E, O = synthetic.CONFIGS[cfg]
        L = p.max_lines_obstacles
        h = oracle.from_params(p, 10)
        s = oracle.sizes(h)
        K = Cn - 1
        assert s["bin"] == Cn * N * (5 * E + R + 5 + 5 * O * L) + K * K * N * 16, cfg
        assert s["cont"] == Cn * N * (12 + 5 * O) + K * K * N * 4, cfg
        assert int(np.asarray(p.possible_region).sum(1).max()) <= 15 and np.all(np.asarray(p.initial_region) >= 1)
        oracle.free(h)


def test_written_dat_reproduces_the_rounded_inputs_exactly(lib, tmp_path):
    """test_hardcoded_data_versus_datfile (test/cplex_wrapper_test.cc:474-505) compares CPPINPUTS and DATFILE runs with
    EXPECT_DOUBLE_EQ: the external-data file written from the (rounded) C++ inputs must read back to the identical
    instance - compared through the LP dump, which prints every coefficient with 17 digits"""
    w = P.CplexWrapper(); w.resetParameters(load_params("cplexmodel_testcase.dat"))
    out = str(tmp_path / "rt.dat")
    assert w.writeDat(out) == 0
    a, b = str(tmp_path / "a.lp"), str(tmp_path / "b.lp")
    assert lib.miqp_solver_export_lp(w._h, a.encode()) == 0
    w2 = P.CplexWrapper(parameterSource=P.ParameterSource.DATFILE); w2.setParameterDatFileAbsolute(out)
    assert w2._push_inputs() == 0 and lib.miqp_solver_export_lp(w2._h, b.encode()) == 0
    assert open(a).read() == open(b).read()


def test_lift_tables_match_a_dense_computation():
    """the response tables behind the bound lifting (host_inst.hpp::lift_tables): Sigma_i = Zu_i H^-1 Zu_i' of every
    triple-integrator chain, rebuilt here from first principles with dense numpy algebra"""
    import planner_miqp_amd as P
    from planner_miqp_amd import synthetic
    p = synthetic.generate("cfg3", 5, gap=0.01, max_time=1.0)
    w = P.CplexWrapper(); w.resetParameters(p)
    T = w.liftTables()
    N, ts = p.NumSteps, p.ts
    A = np.array([[1, ts, ts * ts / 2], [0, 1, ts], [0, 0, 1.0]]); Bv = np.array([ts ** 3 / 6, ts * ts / 2, ts])
    M = N - 1
    for c in range(p.NumCars):
        for ax, names in enumerate((("WEIGHTS_POS_X", "WEIGHTS_VEL_X", "WEIGHTS_ACC_X", "WEIGHTS_JERK_X"), ("WEIGHTS_POS_Y", "WEIGHTS_VEL_Y", "WEIGHTS_ACC_Y", "WEIGHTS_JERK_Y"))):
            wts = np.array([np.asarray(getattr(p, n)).reshape(-1)[c] for n in names])
            Zu = np.zeros((N, 4, M))
            for i in range(N):
                for j in range(min(i, M)):
                    Zu[i, :3, j] = np.linalg.matrix_power(A, i - 1 - j) @ Bv
                if i < M:
                    Zu[i, 3, i] = 1.0
            H = sum(Zu[i].T @ np.diag(2 * wts) @ Zu[i] for i in range(N))
            Hi = np.linalg.inv(H)
            for i in range(N):
                S = Zu[i] @ Hi @ Zu[i].T
                assert np.allclose(T[c, ax, i], S, rtol=1e-8, atol=1e-12), (c, ax, i)
                assert np.allclose(S, S.T) and np.all(np.linalg.eigvalsh(T[c, ax, i]) > -1e-12)
    assert np.all(T[:, :, 0, :3, :] == 0)   # the first state is fixed: only its input responds


def test_bench_strong_scaling_mode_on_two_gloo_ranks():
    """`bench.py --total T --gpus 2` (BASELINE config 4 as written: a fixed set of instances per step split b mod G over the
    ranks): the launcher starts two ranks, every rank takes its share of the SAME seeds, rank 0 prints one JSON line with
    "scaling": "strong".  No GPU here: every solve ends in the loud no-device failure, the partition and the gather are real."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "mini", "--total", "7", "--steps", "2", "--warmup", "0",
                        "--no-cpu", "--time-limit", "1"], capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    o = json.loads(lines[0])
    assert o["scaling"] == "strong" and o["n_gpus"] == 2 and o["steps"] == 2
    assert o["config"]["instances_attempted"] == 14 and [p["attempted"] for p in o["config"]["per_rank"]] == [8, 6]
    assert o["value"] == 0.0 and "no HIP device" in r.stderr


def test_bench_weak_scaling_mode_on_two_gloo_ranks():
    """`bench.py --gpus 2` (the metric's mode: a fixed queue per GPU, instance b of the job on rank b mod G, no data-path
    collective): the two ranks draw DISJOINT seeds that together cover the job's seeds of every step, rank 0 prints one JSON line
    with "scaling": "weak" whose `value` is the job's aggregate.  No GPU here: every solve ends in the loud no-device failure."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MIQP_BENCH_DUMP_SEEDS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "mini", "--batch", "3", "--queue-factor", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu", "--time-limit", "1"], capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    o = json.loads(lines[0])
    assert o["scaling"] == "weak" and o["n_gpus"] == 2 and o["steps"] == 2
    assert o["config"]["instances_attempted"] == 2 * 2 * 6 and [p["attempted"] for p in o["config"]["per_rank"]] == [12, 12]
    assert o["config"]["marshalling_in_timed_region"] is False and o["config"]["result_records_in_timed_region"] is True
    assert o["config"]["collective_backend"] == "gloo" and o["config"]["ranks_seen"] == 2   # what the backend itself saw (an all-reduce of ones); "nccl" = RCCL when every rank has a device
    assert o["value"] == 0.0 and "no HIP device" in r.stderr
    seeds = {}
    for l in r.stderr.splitlines():
        if l.startswith("[bench seeds] "):
            d = json.loads(l[len("[bench seeds] "):]); seeds[d["rank"]] = d["timed"]
    assert sorted(seeds) == [0, 1]
    a, b = set(seeds[0]), set(seeds[1])
    assert len(a) == len(seeds[0]) == 12 and len(b) == 12 and not (a & b)      # disjoint, no repeats
    assert a | b == set(range(12, 36))                                          # warm-up step: seeds 0..11; the two timed steps: 12..35


def test_environment_switches_of_the_shipped_library_are_the_documented_ones():
    """the library reads its environment only through KNOB_P (product) and KNOB_T (tuning builds only), host_inst.hpp: the KNOB_P names
    in the sources == the table of INTEGRATION.md section 5 == the MIQP_* strings of the built .so; no raw getenv("MIQP_...") is left,
    and no KNOB_T name is in the product binary"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ""
    for f in sorted(os.listdir(os.path.join(root, "planner_miqp_amd", "csrc"))):
        src += open(os.path.join(root, "planner_miqp_amd", "csrc", f)).read()
    assert not re.findall(r'getenv\("MIQP_', src), "raw getenv of a MIQP_ switch in the product sources"
    prod = set(re.findall(r'KNOB_P\("(MIQP_[A-Z0-9_]+)"\)', src)); tune = set(re.findall(r'KNOB_T\("(MIQP_[A-Z0-9_]+)"\)', src))
    assert prod and not (prod & tune), prod & tune
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 5. Environment switches"):]
    table = set(re.findall(r"^\| `(MIQP_[A-Z0-9_]+)` \|", sec, re.M))
    assert table == prod, (sorted(table - prod), sorted(prod - table))
    lib = P.library_path()
    if os.path.exists(lib):
        blob = open(lib, "rb").read()
        inbin = set(m.decode() for m in re.findall(rb"MIQP_[A-Z0-9_]+", blob))
        assert prod <= inbin, sorted(prod - inbin)
        assert not (tune & inbin), sorted(tune & inbin)


def test_loading_the_library_asks_for_eight_hardware_queues_unless_the_caller_chose():
    """a round of two cars runs four launches on four streams beside the process's null stream: the library's constructor and
    the Python wrapper set GPU_MAX_HW_QUEUES=8 when it is not set, and leave a value the caller exported alone (checked in fresh processes: the
    HIP runtime reads the variable once)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, ctypes, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None) if sys.argv[1] == '-' else os.environ.__setitem__('GPU_MAX_HW_QUEUES', sys.argv[1]); "
            "ctypes.CDLL(os.path.join(%r, 'planner_miqp_amd', 'libmiqp_gpu.so')); "
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; print((libc.getenv(b'GPU_MAX_HW_QUEUES') or b'').decode())") % (root, root)
    for given, want in (("-", "8"), ("4", "4")):
        out = subprocess.run([sys.executable, "-c", code, given], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-400:]
        assert out.stdout.strip() == want, (given, out.stdout, out.stderr[-200:])
    code2 = "import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); import planner_miqp_amd.wrapper; print(os.environ.get('GPU_MAX_HW_QUEUES'))" % root
    out = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "8", (out.stdout, out.stderr[-300:])

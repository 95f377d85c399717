"""bench.py - MIQP solves/sec to 1% gap on 2-agent x 20-step x 32-region instances (BASELINE.json metric).

A "step" is one pass of the hot path over one QUEUE of synthetic instances: every rank drains its own queue of
independent planning instances with `--batch` of them in flight on its GPU (miqp_solver_solve_stream: a proven instance -
or one that has used up its own 10 s, counted from its admission - hands its slot to the next one of the queue), so the
step time is (queue length) / (solve rate), not the time limit of the hardest instance.  Weak scaling: fixed queue per
GPU, instance b -> rank b mod G, no data-path collective; `value` = instances solved to the gap by all ranks / wall time
of the K timed steps (max over ranks).  `--no-stream` restores the semantics of rounds 1-2 (the whole batch in flight
at once, a step ends when its last instance does).  `--total T` is the strong-scaling mode of BASELINE config 4: a fixed
set of T instances per step (seeds 1000 ...) split b mod G over the ranks.
Inputs are generated and handed to the solver handles before the timed region; the solve call uploads ~22 KB per instance.
"""
import argparse
import json
import numpy as np
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before anything initialises HIP: a round's four launches and the null stream each get a hardware queue (miqp_gpu.hip)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK = 78.6e12  # MI355X FP64 matrix == FP64 vector peak (AMD public spec; half the f32 rate of MI355X_MICROARCH.md)
FP64_PEAK_MEASURED = 71.4e12  # back-to-back v_mfma_f64_16x16x4_f64, 2 waves per SIMD on every CU (profiles/r02_fp64_peak.json)


def f_iter(C, N):
    """algorithmic flops of one interior-point iteration on the stage-banded KKT of one node:
    N stages of size n_s = 8C coupled through n_c = 6C states"""
    ns, nc = 8 * C, 6 * C
    return N * (ns ** 3 / 3.0 + 2 * ns * ns * nc + ns * nc * nc + 4 * ns * ns)


F_ROW = 2 * 6 * (6 + 2)  # sparse assembly per active row and iteration (<= 6 non-zeros per row)
HBM_PEAK = 8.0e12        # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def as_node_units(C, N, fixlen, rows_end, rows_parent, steps):
    """algorithmic work of ONE node relaxation of the dual active-set launch (as_onchip.hip), from the averages of the run:
    bytes = what the node has to move through HBM whatever the implementation: its fix record in, the parent's active set (64 ids) and the
    parent's M (packed triangle over rows_parent rows) in, the solution, the own active set and the own M (rows_end rows) out, the result scalars;
    flops = per step two substitutions with the constant regulator gains (2C chains x N stages x 26 flop each way), the matrix-vector
    product and the symmetric rank-one update on M (4 n^2), the scan for the most violated row (~2 flop per non-zero, ~700 non-zeros), plus the
    multipliers of the start (n^2).  The instance tables the decode reads (28 KB per instance, shared by its nodes) stay in L2."""
    tri = lambda n: n * (n + 1) / 2.0
    nbytes = fixlen + 128 + 8 * tri(rows_parent) + 8 * N * 8 * C + 128 + 8 * tri(rows_end) + 48
    flops = steps * (2 * 2 * (2 * C) * N * 26 + 4 * rows_end ** 2 + 1400) + rows_parent ** 2 * 2
    return nbytes, flops


def cpu_baseline(params_list, gap, time_limit, budget_s=45.0):
    """the CPU oracle (same algorithm class, the checker of the tests - not a tuned CPU solver) on a bounded sample of the same
    workload: one instance per host thread, all started together, each with a limit of `budget_s` (long enough that nearly the
    whole sample finishes, so the rate is not an artefact of the limit); reported: instances proven per second of wall time on
    `cores` threads, the share proven within the device run's own limit (`time_limit`) and within the long limit, and the rate
    of one thread on a smaller sample"""
    import subprocess
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    O = oracle_lib.Oracle(lib)

    def run(params, threads, budget, limit):
        t0 = time.time(); lock = threading.Lock(); state = dict(next=0, times=[], tried=0)

        def worker():
            while True:
                with lock:
                    k = state["next"]
                    if k >= len(params) or time.time() - t0 > budget:
                        return
                    state["next"] = k + 1
                p = params[k]
                h = O.from_params(p, 10)
                a = time.time()
                st, res, pr = O.solve(h, O.dims(p), gap=gap, time_limit=limit)   # ctypes releases the GIL
                b = time.time() - a
                O.free(h)
                with lock:
                    state["tried"] += 1
                    if st == 0 and pr.status in (101, 102):
                        state["times"].append(b)
        ts = [threading.Thread(target=worker) for _ in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return state["times"], state["tried"], time.time() - t0

    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    t1, n1, d1 = run(params_list[:8], 1, 10.0, 2.5)
    sample = params_list[:cores]                      # one instance per thread: nobody waits for a thread
    tc, nc, dc = run(sample, cores, 0.0 if not sample else 1e9, budget_s)
    at_dev = sum(1 for t in tc if t <= time_limit)
    return dict(value=len(tc) / dc if dc > 0 else 0.0, unit="MIQP solves/s", cores=cores, kind="port",
                sample="the first %d instances of rank 0's first timed queue, one per thread on %d threads, %.0f s limit each: %d proven to the gap in %.1f s of wall time "
                       "(%.1f %% of the sample; %d = %.1f %% within the device run's %.0f s limit); one thread, 2.5 s limit: %d of %d in %.1f s"
                       % (nc, cores, budget_s, len(tc), dc, 100.0 * len(tc) / max(1, nc), at_dev, 100.0 * at_dev / max(1, nc), time_limit, len(t1), n1, d1),
                share_proven=len(tc) / max(1, nc), share_proven_at_device_limit=at_dev / max(1, nc), limit_s=budget_s,
                value_at_device_limit=at_dev / min(dc, time_limit) if dc > 0 else 0.0,
                value_1core=len(t1) / d1 if d1 > 0 else 0.0,
                note="the checker of the test suite, not a tuned CPU solver and not CPLEX: a reported baseline, never the target")


def find_cplex():
    """the CPLEX 12.10 the reference links (util/deps.bzl:69-95), if this host has it: interactive optimizer binary or python API"""
    import shutil
    for c in ("/opt/ibm/ILOG/CPLEX_Studio1210/cplex/bin/x86-64_linux/cplex", "/opt/ibm/ILOG/CPLEX_Studio128/cplex/bin/x86-64_linux/cplex"):
        if os.path.exists(c):
            return ("binary", c)
    c = shutil.which("cplex")
    if c:
        return ("binary", c)
    try:
        import cplex  # noqa: F401
        return ("python", "cplex")
    except Exception:
        return None


def cplex_baseline(lp_files, gap, time_limit, budget_s=30.0):
    """CPLEX column of the comparison: each exported LP solved with mipgap / tilim / all threads, solve
    time only; None when CPLEX is not installed (never estimated)"""
    import re
    import subprocess
    found = find_cplex()
    if not found or not lp_files:
        return None
    t0 = time.time(); solved = 0; tried = 0; tsolve = 0.0
    for lp in lp_files:
        if time.time() - t0 > budget_s:
            break
        tried += 1
        if found[0] == "binary":
            r = subprocess.run([found[1], "-c", "read %s" % lp, "set mip tolerances mipgap %g" % gap, "set timelimit %g" % time_limit, "optimize"],
                               capture_output=True, text=True)
            m = re.search(r"Solution time =\s*([0-9.]+) sec", r.stdout)
            ok = "Objective =" in r.stdout and "time limit exceeded" not in r.stdout.lower()
            tsolve += float(m.group(1)) if m else time_limit
        else:
            import cplex
            c = cplex.Cplex(lp); c.set_results_stream(None); c.set_log_stream(None)
            c.parameters.mip.tolerances.mipgap.set(gap); c.parameters.timelimit.set(time_limit)
            a = c.get_time(); c.solve(); tsolve += c.get_time() - a
            ok = c.solution.get_status() in (101, 102)
        solved += int(ok)
    return dict(value=solved / tsolve if tsolve > 0 else 0.0, unit="MIQP solves/s", cores=os.cpu_count(), kind="reference",
                sample="CPLEX (%s) on the first %d exported .lp files, mipgap %g, tilim %g s, default threads: %d to the gap, %.1f s of solve time"
                       % (found[1], tried, gap, time_limit, solved, tsolve))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1280, help="instances IN FLIGHT per GPU (slots of the streaming admission); every instance has its own time limit "
                    "from its admission, so the backlog (in flight x mean work per instance) has to stay below it: 1280 is the largest setting that still proves "
                    ">= 99 %% of the instances in a long stream (--steps 20: 99.3 %%; 1024: 99.7 %%, 1536: 98.8 %%; 128 .. 512: 100 .. 99.9 %% at a lower rate - fewer "
                    "instances offer less parallel work and the hardest 0.5 %% are finished instead of abandoned; 2048: 98.6 %% of a 4-step stream; "
                    "profiles/r03_batch_sweep.json)")
    ap.add_argument("--queue-factor", type=int, default=2, help="instances per GPU and step = queue-factor x batch (the queue one step adds to the stream)")
    ap.add_argument("--no-stream", action="store_true", help="rounds 1-2 semantics: a step is one batch, all of it in flight at once")
    ap.add_argument("--total", type=int, default=0, help="strong scaling (BASELINE config 4): T instances per step in total, seeds 1000 + ..., instance b on rank b mod G")
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--gap", type=float, default=0.01)
    ap.add_argument("--time-limit", type=float, default=10.0, help="max_solution_time per instance (reference default 10 s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--long-extras", action="store_true", help="also the in-flight sweep (256 / 512 in flight at the reference's 10 s limit; a minute more)")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs behind the timed region (all-proven rate, in-flight sweep, one-batch control); they never enter `value`")
    ap.add_argument("--dump-lp", default=None, metavar="DIR", help="write the raw big-M model of every instance of the first timed step as CPLEX .lp (miqp_solver_export_lp) so that a licence holder can fill in the CPLEX column")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU) BEFORE anything
        # touches the GPU, relay rank 0's JSON line and exit with the launcher's code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        # RCCL needs one device per rank; with fewer devices than ranks (a smoke run of the N > 1 path on a small box) the
        # result counters travel over gloo - the solves themselves never use a collective
        dist.init_process_group("nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= world else "gloo")
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if world > ndev and rank == 0:
            print("[bench] %d ranks on %d visible devices: ranks share devices" % (world, ndev), file=sys.stderr)
        local = local % max(1, ndev)
        torch.cuda.set_device(local)
    import planner_miqp_amd as P
    from planner_miqp_amd import synthetic

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    B = a.batch                                   # in flight
    Q = B if a.no_stream else B * max(1, a.queue_factor)   # instances per rank and step
    if a.total > 0:
        # strong scaling: the same T instances whatever the number of ranks, instance b on rank b mod G.  The first TIMED step is
        # BASELINE config 4 as written (seeds 1000 .. 1000 + T - 1), later timed steps follow on; the warm-up steps use the seeds after them
        def seeds_of(step):
            k = step - a.warmup if step >= a.warmup else a.steps + step
            return [1000 + k * a.total + b for b in range(a.total) if b % world == rank]
    else:
        # weak scaling: seeds rank*Q .. rank*Q+Q-1 of step s are offset by s*world*Q
        def seeds_of(step):
            return [(step * world + rank) * Q + k for k in range(Q)]

    def make_batch(step):
        ps = [synthetic.generate(a.config, sd, gap=a.gap, max_time=a.time_limit) for sd in seeds_of(step)]
        ws = []
        for p in ps:
            w = P.CplexWrapper(device=local); w.resetParameters(p); ws.append(w)
        P.prepare_batch(ws)   # parameters into the solver handles (host side; the device upload is part of the timed solve call)
        return ps, ws

    batches = [make_batch(s) for s in range(a.warmup + a.steps)]
    if os.environ.get("MIQP_BENCH_DUMP_SEEDS"):   # (tests: which seeds this rank solves in its timed steps)
        print("[bench seeds] " + json.dumps(dict(rank=rank, timed=[sd for st_ in range(a.warmup, a.warmup + a.steps) for sd in seeds_of(st_)])), file=sys.stderr, flush=True)
    infl = None if a.no_stream else B
    lp_files = []
    if a.dump_lp and rank == 0:
        os.makedirs(a.dump_lp, exist_ok=True)
        L = P.load_library()
        for k, w in enumerate(batches[a.warmup][1]):
            if True:
                f = os.path.join(a.dump_lp, "%s_seed%d.lp" % (a.config, seeds_of(a.warmup)[k]))
                if L.miqp_solver_export_lp(w._h, f.encode()) == 0:
                    lp_files.append(f)
    if a.warmup > 0:
        wws = [w for s in range(a.warmup) for w in batches[s][1]]
        if a.no_stream:
            for s in range(a.warmup):
                P.solve_batch(batches[s][1], prepared=True)
        elif wws:
            P.solve_batch(wws, inflight=infl, prepared=True)
    sync()
    t0 = time.time()
    solved = 0; attempted = 0; ipm_s = 0.0; launches = 0; iters = 0; rowit = 0; nodes = 0; asx = [0.0] * 8; lat = []; results_s = 0.0; nrec = 0; stream_info = None; stream_ws = None
    # streaming: the queues of the K timed steps are drained as ONE stream (a step = its queue of instances; no idle tail
    # between steps: the slots freed by the last instances of one queue go to the first of the next); --no-stream: step by step
    timed = [batches[s] for s in range(a.warmup, a.warmup + a.steps)]
    if not a.no_stream:
        timed = [([p for ps, _ in timed for p in ps], [w for _, ws in timed for w in ws])]
    for ps, ws in timed:
        sts = P.solve_batch(ws, inflight=infl, prepared=True) if ws else []
        attempted += len(ws)
        if not ws:
            continue
        # the RawResults record of every solved instance (collectRawResults runs inside the reference's callCplex): built on the
        # host threads inside the library, inside the timed region
        tr = time.time(); nrec += P.materialize_results(ws, int(os.environ.get("MIQP_BENCH_MAT_THREADS", "0"))); results_s += time.time() - tr
        for w, st in zip(ws, sts):
            pr = w.getSolutionProperties()
            ok = st == P.OptimizationStatus.SUCCESS and pr.status in (101, 102)
            solved += int(ok)
            if ok:
                lat.append(pr.time)   # seconds from the instance's admission to its proof
        tm = ws[0].lastTiming()
        ipm_s += tm["ipm_s"]; launches += tm["ipm_launches"]; iters += tm["ipm_iters"]; rowit += tm["row_iters"]; nodes += tm["nodes"]
        for k_, key_ in enumerate(("as_nodes", "as_steps", "as_unfinished", "as_drops", "as_rows_end", "as_rows_parent", "std_launch_s", "std_launches")):
            asx[k_] += tm[key_]
        stream_ws = (ws, sts)
    sync()
    dt = time.time() - t0
    if not a.no_stream and stream_ws is not None:   # (outside the timed region) when the stream's last instance was admitted, and what had been proven by then
        ws, sts = stream_ws
        adm = [w.lastAdmission() for w in ws]; t_last = max(adm)
        prs = [w.getSolutionProperties() for w in ws]
        done_by = sum(1 for x, pr, st in zip(adm, prs, sts) if st == P.OptimizationStatus.SUCCESS and pr.status in (101, 102) and x + pr.time <= t_last)
        stream_info = dict(last_admission_s=float(t_last), proven_by_then=int(done_by), solves_per_s_with_backlog=(done_by / t_last if t_last > 0 else None),
                           note="the timed stream on rank 0: its last instance is admitted at last_admission_s, the rest of ms_per_step x steps is the end effect of the finite queue - the stream "
                                "waits for the instances admitted last, up to the per-instance limit, with most slots empty (a service whose queue stays filled runs at solves_per_s_with_backlog; "
                                "reported beside `value`, never as it; meaningful only for a stream that lasts several times the per-instance limit - in a short one the slots are still filling up with the slow instances when the queue runs out)")
    from planner_miqp_amd.sharding import gather_counts
    dev = torch.device("cuda", local) if (world > 1 and torch.cuda.is_available() and torch.cuda.device_count() >= world) else None
    g = gather_counts([dt, solved, attempted, ipm_s, launches, iters, rowit, nodes], dev)
    ranks_seen = 1
    if world > 1:   # what the collective backend itself saw: an all-reduce(sum) of one per rank
        one = torch.ones(1, dtype=torch.int64, device=dev if dev is not None else "cpu")
        dist.all_reduce(one); ranks_seen = int(one.item())

    def leg(ws_, inflight_, marshal=False, limit=None):
        """one extra leg on rank 0 behind the timed region: a queue drained with `inflight_` in flight (None: one batch, all in flight);
        returns (proven, attempted, seconds, nodes, slowest instance); marshal=True puts parameter marshalling inside its timer;
        limit: the per-instance time limit of this leg (miqp_solver_override_settings; the gap stays)"""
        L_ = P.load_library()
        for w in ws_:
            L_.miqp_solver_override_settings(w._h, float(limit if limit is not None else a.time_limit), float(a.gap))
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        a0 = time.time()
        if marshal:
            P.prepare_batch(ws_)
            for w in ws_:
                L_.miqp_solver_override_settings(w._h, float(limit if limit is not None else a.time_limit), float(a.gap))
        st_ = P.solve_batch(ws_, inflight=inflight_, prepared=True)
        P.materialize_results(ws_)
        d_ = time.time() - a0
        tm_ = ws_[0].lastTiming()
        if tm_["context_built"]:
            d_ -= tm_["context_s"]   # a leg with another number of slots rebuilds the device context (seconds of hipMalloc): a service keeps its shape, the timed region above is warmed up
        prs = [w.getSolutionProperties() for w in ws_]
        okv_ = [t_ == P.OptimizationStatus.SUCCESS and pr_.status in (101, 102) for pr_, t_ in zip(prs, st_)]
        leg.times = [pr_.time for pr_, o_ in zip(prs, okv_) if o_]   # seconds from admission to proof of the instances this leg proved
        # drain rate while the queue still had a backlog: the instances proven by the time the LAST instance of the queue was admitted, per second
        # of that time - a service that keeps its queue filled runs in this state; what follows the last admission is the end effect of a finite
        # queue (the pass then waits for its slowest instances with most slots empty)
        adm_ = [w.lastAdmission() for w in ws_]
        t_last_ = max(adm_) if adm_ else 0.0
        leg.steady = (sum(1 for a_, pr_, o_ in zip(adm_, prs, okv_) if o_ and a_ + pr_.time <= t_last_) / t_last_, t_last_) if t_last_ > 0 else (None, 0.0)
        return sum(okv_), len(ws_), d_, int(tm_["nodes"]), max(pr_.time for pr_ in prs)

    def pct(v):
        return dict(p50=float(np.percentile(v, 50)), p95=float(np.percentile(v, 95)), p99=float(np.percentile(v, 99)), max=float(max(v))) if v else None

    extras = None
    if rank == 0 and world == 1 and not a.no_extras and not a.no_stream and a.total <= 0:
        # Legs behind the timed region (rank 0, single GPU; they re-solve instances of the timed queues, results are discarded):
        #  * all_proven_at_bench_in_flight: the whole timed stream again, same in-flight setting, limit 60 s instead of 10 s;
        #  * time_to_prove_all: the head of the stream (2048 instances) at 256 in flight, limit 60 s - ends with its slowest instance;
        #  * in_flight_sweep (--long-extras): the same head at 256 / 512 in flight with the reference's 10 s limit;
        #  * one_batch_control: the semantics of rounds 1-2 with marshalling and result records inside the timer.
        pool = [w for _, ws in timed for w in ws]
        qn = min(len(pool), 2048)
        # (1) the knob-free THROUGHPUT: the whole timed stream once more at the bench's own in-flight setting with the per-instance limit
        # lifted to 60 s - nothing is abandoned at 10 s; solves/s over the stream, the share proven, and the time from admission to proof
        ok_, n_, d_, nd_, mx_ = leg(pool, B, limit=60.0)
        aph_ = dict(in_flight=B, queue=n_, time_limit_s=60.0, solves_per_s=ok_ / d_, proven_share=ok_ / n_, seconds=d_, nodes_per_instance=nd_ / n_, slowest_instance_s=mx_,
                    time_to_proof_s=pct(leg.times), solves_per_s_with_backlog=leg.steady[0], last_admission_s=leg.steady[1],
                    note="solves_per_s: the whole pass, to the end of its slowest instance; solves_per_s_with_backlog: instances proven until the last one of the queue was admitted / that time (the drain rate while a backlog exists, the end effect of the finite queue left out - meaningful only for a stream that lasts several times the slowest instance: the driver's 20 steps, not the 4-step default, whose last admission comes after 2 s)")
        # (2) time to prove ALL of a 2048-instance queue at 256 in flight (60 s limit): ends with its slowest instance - a latency of the hardest
        # instance of the queue, not a throughput (it was reported as `all_proven` / `value_all_proven` in round 4)
        ok_, n_, d_, nd_, mx_ = leg(pool[:qn], 256, limit=60.0)
        ap_ = dict(in_flight=256, queue=n_, time_limit_s=60.0, solves_per_s=ok_ / d_, proven_share=ok_ / n_, seconds=d_, nodes_per_instance=nd_ / n_, slowest_instance_s=mx_,
                   time_to_proof_s=pct(leg.times))
        sweep = []
        for infl_ in ((256, 512) if a.long_extras else ()):
            if 4 * infl_ > len(pool):
                break
            ok_, n_, d_, nd_, mx_ = leg(pool[:qn], infl_)
            sweep.append(dict(in_flight=infl_, queue=n_, time_limit_s=a.time_limit, solves_per_s=ok_ / d_, proven_share=ok_ / n_, seconds=d_, nodes_per_instance=nd_ / n_, slowest_instance_s=mx_))
        nb_ = min(1024, len(pool))
        ok_, n_, d_, nd_, mx_ = leg(pool[:nb_], None, marshal=True)
        extras = dict(in_flight_sweep=sweep, time_to_prove_all=ap_, all_proven_at_bench_in_flight=aph_,
                      one_batch_control=dict(instances=n_, solves_per_s=ok_ / d_, proven_share=ok_ / n_, seconds=d_, slowest_instance_s=mx_,
                                             note="--no-stream semantics (one batch, all in flight, ends with its last instance); parameter marshalling, device upload and result records inside the timer"))
        for w in pool:
            P.load_library().miqp_solver_override_settings(w._h, float(a.time_limit), float(a.gap))

    if rank == 0:
        T = max(x[0] for x in g)
        tot_solved = sum(x[1] for x in g); tot_att = sum(x[2] for x in g)
        Cc, N = synthetic.CONFIGS[a.config][0], synthetic.CONFIGS[a.config][1]
        # roofline of the dominant kernel, rank 0.  Two cars with the active-set launches (the default since round 6): the standard launch
        # as_onchip_kernel<2,10,128>, timed alone with HIP events on the solver stream; its algorithmic unit is one node relaxation
        # (as_node_units); by its arithmetic intensity (~3 flop per byte against a machine balance of ~10) the bounding roofline is HBM.
        # Otherwise (one, three, four cars; MIQP_AS=0): the interior point launches, algorithmic flops / HIP-event time of the launch group
        as_nodes, as_steps = asx[0], asx[1]
        ipm_iters_only = max(0.0, iters - as_steps)           # (the steps of the active-set launches are part of NrIterations, not interior point iterations)
        flops = ipm_iters_only * f_iter(Cc, N) + rowit * F_ROW
        use_as = as_nodes > 0 and asx[6] > 0 and asx[7] > 0
        # L2<->fabric bytes per launch: not measurable inside this process; the figure of the committed rocprofv3 --pmc passes of the same configuration
        traffic = None; traffic_note = None
        tj = next((q for q in (os.path.join(ROOT, "profiles", "r%02d_traffic.json" % r) for r in (6, 5, 4, 3, 2)) if os.path.exists(q)), "")
        if os.path.exists(tj) and a.config == "cfg3":
            tjd = json.load(open(tj))
            dk_ = tjd.get("dominant_kernel_bytes_per_launch_corrected") if use_as else None   # (per launch of the dominant kernel, like `achieved`; files of rounds <= 5: the interior point launches of a round)
            traffic = dk_ if dk_ is not None else tjd.get("bytes_per_round_corrected")
            traffic_note = "%s%s [file %s, from %s]" % ("L2<->fabric bytes per launch of as_onchip_kernel<2,10,128> alone; " if dk_ is not None else "", tjd.get("note"), os.path.basename(tj), tjd.get("source"))
        if use_as:
            fixlen = 16 * ((Cc * N * (1 + 5 + 5 * synthetic.CONFIGS[a.config][4]) + (Cc * (Cc - 1) // 2) * N * 8 + Cc * N * 2 + 15) // 16)   # bytes of a node's fix record
            n_end = asx[4] / as_nodes; n_par = asx[5] / as_nodes; st_ = as_steps / as_nodes
            nb_, fl_ = as_node_units(Cc, N, fixlen, n_end, n_par, st_)
            std_s = asx[6]; std_l = asx[7]
            # (nodes of the STANDARD launch: all active-set nodes but the ones of the larger block, which are not counted apart; the share of the
            # larger block is 5-7 % of the nodes - the figure below attributes every node to the standard launch and is an upper bound by that much)
            ach = as_nodes * nb_ / std_s
            roof = dict(bound="hbm", achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=ach / HBM_PEAK, traffic=traffic, traffic_note=traffic_note,
                        kernel="as_onchip_kernel<2,10,128>: the dual active-set solves of the ordinary nodes of a B&B round (the standard launch of the round's four concurrent launches; "
                               "HIP events on the solver stream from the start of the launch group to the end of this kernel)",
                        launches=int(std_l), avg_launch_ms=1e3 * std_s / max(1.0, std_l), bytes_per_launch=as_nodes * nb_ / max(1.0, std_l),
                        algorithmic_bytes_per_node=nb_, algorithmic_flops_per_node=fl_, flops_achieved_tflops=as_nodes * fl_ / std_s / 1e12,
                        nodes_per_launch=as_nodes / max(1.0, std_l), nodes_per_s=as_nodes / std_s,
                        steps_per_node=st_, active_rows_per_node=n_end, rows_from_parent_per_node=n_par, unfinished_nodes=int(asx[2]),
                        launch_group_ms=1e3 * ipm_s / max(1, launches), launch_groups=int(launches),
                        interior_point_beside_it=dict(iterations=int(ipm_iters_only), tflops=(flops / ipm_s / 1e12 if ipm_s > 0 else 0.0),
                                                      note="the larger interior point variant and the memory-backed kernel on their own streams: the nodes the active-set launches leave to them, and the polish"),
                        note="the node relaxation is no dense contraction any more: the Hessian of a node QP is the objective's alone, so H^-1 is a substitution with constant gains over 2C scalar chains and "
                             "the only dense object is the inverse of the active rows' Schur complement (n ~ 35); per node ~0.1 Mflop and ~17 KB instead of 12.2 interior point iterations x 253 kflop. "
                             "By intensity (flop per byte against 78.6 TFLOP/s : 8 TB/s) the roofline that bounds the kernel is HBM, and it is far from it: the kernel is bound by latency - the dependent "
                             "L2 loads of the row decode and LDS round trips of one wavefront per node (DESIGN.md 6); the figure of merit is nodes_per_s")
        else:
            ach = flops / ipm_s if ipm_s > 0 else 0.0
            roof = dict(bound="fp64_vector", achieved=ach / 1e12, peak=FP64_PEAK / 1e12, unit="TFLOP/s", frac=ach / FP64_PEAK, traffic=traffic,
                        traffic_note=traffic_note, peak_measured=FP64_PEAK_MEASURED / 1e12, frac_of_measured_peak=ach / FP64_PEAK_MEASURED,
                        kernel="the interior point launches of a B&B round (HIP events around the group on the solver stream)",
                        launches=int(launches), avg_launch_ms=1e3 * ipm_s / max(1, launches), flops_per_launch=flops / max(1, launches),
                        nodes_per_s=(nodes / ipm_s if ipm_s > 0 else 0.0),
                        note="FP64 vector roofline: on gfx950 the FP64 MFMA peak equals the FP64 VALU peak, and MFMA instructions are < 1 % of these kernels' vector instructions")
        out = dict(metric="MIQP solves/sec to 1% gap, 2-agent x 20-step x 32-region", value=tot_solved / T, unit="MIQP solves/s",
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=1e3 * T / a.steps, higher_is_better=True, scaling="strong" if a.total > 0 else "weak",
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload="%s: %d cars x %d steps x %d regions, %d env pieces, %d obstacles; %s, gap %g, time limit %g s per instance%s"
                               % ((a.config,) + synthetic.CONFIGS[a.config] + (
                                   ("%d instances per step in total, split b mod G" % a.total) if a.total > 0 else ("queue of %d instances per GPU and step, the %d timed steps drained as one stream" % (Q, a.steps)),
                                   a.gap, a.time_limit,
                                   " (whole batch in flight, the step ends with its last instance)" if a.no_stream else (" from its admission, %d in flight per GPU (streaming admission)" % B))),
                               in_flight=B, queue_per_gpu_and_step=(None if a.total > 0 else Q), streaming=not a.no_stream,
                               marshalling_in_timed_region=False, result_records_in_timed_region=True,
                               result_records_built=int(nrec), result_records_seconds_rank0=round(results_s, 3),
                               instances_attempted=int(tot_att), instances_solved_to_gap=int(tot_solved),
                               per_rank=[dict(rank=k, seconds=round(x[0], 3), solved=int(x[1]), attempted=int(x[2]), bnb_nodes=int(x[7])) for k, x in enumerate(g)],
                               bnb_nodes=int(sum(x[7] for x in g)), ipm_iterations=int(sum(x[5] for x in g)), active_set_steps_rank0=int(asx[1]), active_set_nodes_rank0=int(asx[0]),
                               solve_latency_s_rank0=dict(p50=float(np.percentile(lat, 50)), p95=float(np.percentile(lat, 95)), p99=float(np.percentile(lat, 99)), max=float(max(lat))) if lat else None,
                               timed_stream_rank0=stream_info,
                               collective_backend=(dist.get_backend() if world > 1 else None), ranks_seen=int(ranks_seen)),
                   roofline=roof)
        out["schema"] = 6   # (round 6: roofline describes the active-set launch where it runs; `time_to_prove_all` is what rounds <= 4 called `all_proven`)
        if extras:
            # `value` depends on the instances in flight (more in flight = the hardest instances are abandoned at their limit sooner);
            # the knob-free figure is the rate at a setting that proves EVERY instance of its queue
            # `value` depends on the instances in flight (more in flight = the hardest instances are abandoned at their own 10 s sooner).  The
            # knob-free figure: the same stream at the same setting with the limit lifted to 60 s - `value_limit_lifted` (solves/s of that pass,
            # with the share it proved beside it); `value_all_proven` only when that pass proved every instance
            aph_ = extras["all_proven_at_bench_in_flight"]
            out["value_limit_lifted"] = aph_["solves_per_s"]; out["proven_share_limit_lifted"] = aph_["proven_share"]
            out["value_all_proven"] = aph_["solves_per_s"] if aph_["proven_share"] >= 1.0 else None
            out["all_proven_at_bench_in_flight"] = aph_; out["time_to_prove_all"] = extras["time_to_prove_all"]
            out["all_proven"] = extras["time_to_prove_all"]   # (alias: the key of rounds <= 4 for the same leg - 2048 instances at 256 in flight, limit 60 s)
            out["in_flight_sweep"] = extras["in_flight_sweep"]; out["one_batch_control"] = extras["one_batch_control"]
        out["proven_share"] = tot_solved / max(1, tot_att)
        if not a.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(batches[a.warmup][0], a.gap, a.time_limit)
            cpx = cplex_baseline(lp_files, a.gap, a.time_limit) if lp_files else None
            out["cplex_baseline"] = cpx if cpx else ("CPLEX not available on this host" + ("" if lp_files else " (run with --dump-lp DIR to export the instances)"))
        elif not a.no_cpu:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""gpurun_out/final (tools/final_profiles.sh) + gpurun_out/final_numbers (tools/final_numbers.sh) -> profiles/r03_*: copies the
rocprofv3 summaries and writes profiles/r03_traffic.json (L2 <-> fabric bytes per launch of the two interior point kernels,
FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) - the file bench.py reads `roofline.traffic` from."""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, N, P = os.path.join(R, "gpurun_out", "final"), os.path.join(R, "gpurun_out", "final_numbers"), os.path.join(R, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
for src, dst in (("kernel_stats.csv", "final_kernel_stats.csv"), ("trace_bench.json", "final_trace_bench.json"), ("traffic_raw.json", "final_pmc_traffic.json"), ("sq_raw.json", "final_pmc_sq.json")):
    shutil.copy(os.path.join(F, src), os.path.join(P, "%s_%s" % (tag, dst)))
t = json.load(open(os.path.join(F, "traffic_raw.json")))
def kern(sub):
    k = [v for n, v in t.items() if sub in n][0]
    return dict(fetch_bytes_raw=k["FETCH_SIZE"]["bytes_per_launch"], fetch_bytes_corrected=2.0 * k["FETCH_SIZE"]["bytes_per_launch"], write_bytes=k["WRITE_SIZE"]["bytes_per_launch"], launches=k["FETCH_SIZE"]["launches"])
oc, ob, ov = kern("ipm_onchip_kernel<2, 10, 0, 128>"), kern("ipm_onchip_kernel<2, 10, 0, 320>"), kern("ipm_kernel<2")
out = dict(bytes_per_round_corrected=sum(k["fetch_bytes_corrected"] + k["write_bytes"] for k in (oc, ob, ov)), onchip_kernel=oc, onchip_kernel_larger_variant=ob, overflow_kernel=ov,
           note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no tracing) of `python3 bench.py --steps 1 --warmup 0 --no-cpu --time-limit 3` (the bench configuration: cfg3, 1024 in flight, queue of 2048; the 3 s limit keeps the counter passes short); counters are KiB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; both count L2<->fabric requests (Infinity-Cache hits included), per launch averaged over all launches of the pass; one B&B round = one launch of ipm_onchip_kernel<2,10,0,128>, beside it on a second stream one of its larger variant <2,10,0,320> (rounding probes, marked large nodes) and one of ipm_kernel<2,64> (what that one cannot hold)",
           source="profiles/%s_final_pmc_traffic.json (tools/final_profiles.sh, tools/pmc_traffic.py, tools/summarize_profiles.py)" % tag)
json.dump(out, open(os.path.join(P, "%s_traffic.json" % tag), "w"), indent=1)
if os.path.isdir(N):
    shutil.copy(os.path.join(N, "bench_default.json"), os.path.join(P, "%s_final_bench.json" % tag))
    shutil.copy(os.path.join(N, "batch_sweep.jsonl"), os.path.join(P, "%s_batch_sweep.json" % tag))
    shutil.copy(os.path.join(N, "single_latency.txt"), os.path.join(P, "%s_single_latency.txt" % tag))
print(json.dumps({k: (round(v / 1e9, 2) if isinstance(v, float) else v) for k, v in out.items() if k == "bytes_per_round_corrected"}))

"""Time to gap of ONE instance solved plainly and with its tree split over W ranks (C1) - on a one-GPU box the
ranks are processes that share the device, so this measures what the split costs (root partition, one all-reduce per round over
gloo, W contexts on one device), not what W devices gain.  python tools/split_timing.py [W]   (GPU only; launches itself)"""
import json, os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("cfg3", 307, 0.01), ("cfg3", 118, 0.01), ("cfg3", 662, 0.01), ("cfg5", 0, 0.01), ("cfg5", 3, 0.01), ("cfg5", 5, 0.01), ("cfg5", 9, 0.01)]

def worker():
    sys.path.insert(0, ROOT)
    import torch, torch.distributed as dist
    import planner_miqp_amd as P
    from planner_miqp_amd import synthetic, sharding
    dist.init_process_group("gloo")
    r, w = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(r % torch.cuda.device_count())
    ex = sharding.torch_exchange()
    out = []
    for cfg, seed, gap in CASES:
        x = P.CplexWrapper(device=torch.cuda.current_device()); x.resetParameters(synthetic.generate(cfg, seed, gap=gap, max_time=10))
        sharding.split_solve(x, ex)   # builds the context of this shape
        dist.barrier(); t = time.time()
        st = sharding.split_solve(x, ex)
        dt = time.time() - t
        pr = x.getSolutionProperties()
        out.append(dict(cfg=cfg, seed=seed, status=int(st), cplex_status=int(pr.status), seconds=dt, objective=pr.objective, gap=pr.gap, nodes=int(pr.nodes)))
    allo = [None] * w
    dist.all_gather_object(allo, out)
    if r == 0:
        print("SPLIT_JSON " + json.dumps(allo))
    dist.destroy_process_group()

def launch(W):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SPLIT_WORKER="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % W, "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.abspath(__file__)], capture_output=True, text=True, env=env, timeout=1200)
    line = [l for l in out.stdout.splitlines() if l.startswith("SPLIT_JSON ")]
    if not line:
        print(out.stdout[-2000:], out.stderr[-2000:]); return None
    return json.loads(line[0][len("SPLIT_JSON "):])

if __name__ == "__main__":
    if os.environ.get("SPLIT_WORKER"):
        worker()
    else:
        W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        one, many = launch(1), launch(W)
        for k, (cfg, seed, gap) in enumerate(CASES):
            a = one[0][k] if one else None; b = many[0][k] if many else None
            nodes_w = sum(rk[k]["nodes"] for rk in many) if many else None
            print(json.dumps(dict(cfg=cfg, seed=seed, gap=gap, one_rank=dict(seconds=round(a["seconds"], 3), status=a["cplex_status"], gap=a["gap"], nodes=a["nodes"]) if a else None,
                                  ranks=W, split=dict(seconds=round(b["seconds"], 3), status=b["cplex_status"], gap=b["gap"], nodes_all_ranks=nodes_w) if b else None)))

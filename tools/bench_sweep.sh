#!/bin/bash
# headline against the number of instances in flight (GPU): one summary line per bench run -> profiles/r03_batch_sweep.json
S=${STEPS:-4}
for b in 128 256 512 1024 1280 1536 2048; do
  python bench.py --batch $b --steps $S --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print(json.dumps(dict(in_flight=$b, queue_per_step=c['queue_per_gpu_and_step'], steps=d['steps'], solves_per_s=round(d['value'],1), solved=c['instances_solved_to_gap'], attempted=c['instances_attempted'], proven_share=round(c['instances_solved_to_gap']/c['instances_attempted'],4), bnb_nodes=c['bnb_nodes'], nodes_per_instance=round(c['bnb_nodes']/c['instances_attempted']), ms_per_step=round(d['ms_per_step'],1), latency=c['solve_latency_s_rank0'], roofline_frac=round(d['roofline']['frac'],4), avg_launch_ms=round(d['roofline']['avg_launch_ms'],2))))"
done

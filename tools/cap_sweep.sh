#!/bin/bash
# share cap x in-flight x queue length (GPU): one line per bench run
for cap in 512 1024 2048 4096; do for b in 256 1024; do for qf in 4 8; do
  MIQP_SHARE_CAP=$cap python bench.py --batch $b --queue-factor $qf --steps 1 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('cap $cap inflight $b qf $qf: %.1f solves/s, %d/%d, %.2fM nodes, step %.2f s, p95 %.2f max %.2f'%(d['value'],c['instances_solved_to_gap'],c['instances_attempted'],c['bnb_nodes']/1e6,d['ms_per_step']/1e3,c['solve_latency_s_rank0']['p95'],c['solve_latency_s_rank0']['max']))"
done; done; done

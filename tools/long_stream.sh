#!/bin/bash
# the driver's 20-step stream at a given in-flight setting and share policy: value and share proven.  tools/long_stream.sh BATCH [ENV=VAL ...]
b=$1; shift
env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --batch $b --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('in flight $b $*: value %.1f proven %d / %d = %.4f nodes/inst %d p95 %.2f' % (d['value'], c['instances_solved_to_gap'], c['instances_attempted'], c['instances_solved_to_gap']/c['instances_attempted'], c['bnb_nodes']/c['instances_attempted'], c['solve_latency_s_rank0']['p95']))"

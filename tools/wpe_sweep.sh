#!/bin/bash
# register allocation of the memory-backed kernel (wavefronts per SIMD it is compiled for): bench value and single-solve latency
for v in "" wpe2 wpe1; do
  if [ -n "$v" ]; then export MIQP_GPU_LIB=$PWD/tools/_build/libmiqp_gpu_$v.so; fi
  echo "== ${v:-default (3)}"
  python bench.py --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s roofline %.3f launch ms %.2f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), d['roofline']['frac'], d['roofline'].get('avg_launch_ms', 0)))
"
  python tools/single_latency.py 96 0.1 | tail -n 1
done

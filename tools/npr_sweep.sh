#!/bin/bash
# nodes per round (batch capacity = in flight x MIQP_NPR) against the bench value
run() { echo "== $*"; env $1 python bench.py ${@:2} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('value %.1f ms/step %.0f proven %s/%s roofline %.3f' % (d['value'], d['ms_per_step'], c.get('instances_solved_to_gap'), c.get('instances_attempted'), d['roofline']['frac']))
"; }
run MIQP_NPR=16
run MIQP_NPR=48
run MIQP_NPR=64
run MIQP_NPR=32 --batch 2048

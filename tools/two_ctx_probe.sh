#!/bin/bash
# does the memory-backed launch of one queue hide behind the on-chip launch of another?  The same queue (seeds 0..2047, no instance
# near its time limit, 512 in flight, 32 nodes per instance and round) alone, and twice / three times at once on the same device
one() { python tools/stream_check.py $1 $2 $3 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('Q %d inflight %d: solved %d, rounds loop %.2f s, wall %.2f s, nodes %d' % (d['Q'], d['inflight'], d['solved'], d['timing']['solve_s'], d['seconds'], d['nodes']))"; }
export MIQP_NPR=32
echo "== alone"; one 2048 512 0
echo "== two at once"
one 2048 512 0 & p1=$!
one 2048 512 0 & p2=$!
wait $p1; wait $p2
echo "== three at once"
one 2048 512 0 & p1=$!
one 2048 512 0 & p2=$!
one 2048 512 0 & p3=$!
wait $p1; wait $p2; wait $p3
echo "== alone, 1024 in flight"; one 2048 1024 0

#!/bin/bash
# streaming policy sweep (GPU): MIQP_SHARE_CAP x MIQP_BASE_TAKE on queues of cfg3 instances; one JSON line per run
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for cfgq in "1024 256" "4096 256" "4096 1024"; do
  for pol in "4096 8" "2048 8" "8192 8" "4096 32"; do
    set -- $pol
    echo "== Q/inflight $cfgq share_cap $1 base $2"
    MIQP_SHARE_CAP=$1 MIQP_BASE_TAKE=$2 python tools/stream_check.py $cfgq 2>&1 | tail -n 1 | cut -c1-420
  done
done

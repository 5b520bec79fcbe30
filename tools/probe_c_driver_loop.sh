#!/bin/bash
# Run examples/c_driver N times on the GPU box, each under its own time limit with line-buffered output and RCCL's own init
# log (NCCL_DEBUG=INFO): a run that does not come back shows how far it got and, if rocgdb is there, where its threads stand.
# usage: tools/probe_c_driver_loop.sh [N=25] [per-run limit, s = 40] [extra env assignments, e.g. NCCL_IB_DISABLE=1]
cd $GRAFT_REPO_ROOT
N=${1:-25}; T=${2:-40}; shift 2
gcc -std=c99 -I include examples/c_driver.c -o /tmp/c_driver_loop -L raft_rs_amd -lraftgroups -Wl,-rpath,$GRAFT_REPO_ROOT/raft_rs_amd || exit 1
for i in $(seq 1 $N); do
  s=$(date +%s)
  env NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,NET,BOOTSTRAP,ENV "$@" stdbuf -oL -eL /tmp/c_driver_loop > /tmp/c_driver_loop.out 2>&1 &
  pid=$!
  while kill -0 $pid 2>/dev/null && [ $(( $(date +%s) - s )) -lt $T ]; do sleep 1; done
  if kill -0 $pid 2>/dev/null; then
    echo "run $i: still running after $T s; output so far:"; tail -30 /tmp/c_driver_loop.out | cut -c1-220
    cpid=$(pgrep -P $pid | head -1); [ -z "$cpid" ] && cpid=$pid
    if command -v rocgdb > /dev/null; then
      echo "---- threads (rocgdb):"; timeout 60 rocgdb -p $cpid -batch -ex "thread apply all bt 12" 2>&1 | grep -v "^\[New\|^warning\|^Reading\|^Loaded" | tail -60 | cut -c1-200
    fi
    kill -9 $cpid $pid 2>/dev/null
    break
  fi
  wait $pid; rc=$?
  echo "run $i: rc $rc in $(( $(date +%s) - s )) s"
  if [ $rc -ne 0 ]; then tail -25 /tmp/c_driver_loop.out; break; fi
done

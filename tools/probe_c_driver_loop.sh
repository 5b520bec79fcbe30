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
    echo "run $i: still running after $T s; output so far:"; tail -5 /tmp/c_driver_loop.out | cut -c1-220
    cpid=$(pgrep -P $pid | head -1); [ -z "$cpid" ] && cpid=$pid
    # what is it doing? (ptrace is not permitted on the box: /proc instead) -- twice, 10 s apart, then wait for it up to 10 min in all
    for k in 1 2; do
      echo "---- /proc/$cpid: $(grep State /proc/$cpid/status | tr -s '\t ' ' ') wchan=$(cat /proc/$cpid/wchan 2>/dev/null) rccl maps=$(grep -c rccl /proc/$cpid/maps)"
      grep -E "rchar|read_bytes" /proc/$cpid/io | tr '\n' ' '; echo
      for t in /proc/$cpid/task/*; do echo "  thread $(basename $t): $(cat $t/comm) $(grep State $t/status | tr -s '\t ' ' ') wchan=$(cat $t/wchan 2>/dev/null)"; done | head -12
      sleep 10
    done
    while kill -0 $pid 2>/dev/null && [ $(( $(date +%s) - s )) -lt 600 ]; do sleep 2; done
    if kill -0 $pid 2>/dev/null; then echo "run $i: NOT back after 600 s"; kill -9 $cpid $pid 2>/dev/null; break; fi
    wait $pid; echo "run $i: came back with rc $? after $(( $(date +%s) - s )) s"; tail -12 /tmp/c_driver_loop.out | cut -c1-200
    continue
  fi
  wait $pid; rc=$?
  echo "run $i: rc $rc in $(( $(date +%s) - s )) s"
  if [ $rc -ne 0 ]; then tail -25 /tmp/c_driver_loop.out; break; fi
done

#!/bin/bash
# Run examples/c_driver N times on the GPU box, each under its own timeout with line-buffered output: a run that hangs shows
# how far it got. usage: tools/probe_c_driver_loop.sh [N=25] [per-run timeout, s = 40]
cd $GRAFT_REPO_ROOT
N=${1:-25}; T=${2:-40}
gcc -std=c99 -I include examples/c_driver.c -o /tmp/c_driver_loop -L raft_rs_amd -lraftgroups -Wl,-rpath,$GRAFT_REPO_ROOT/raft_rs_amd || exit 1
for i in $(seq 1 $N); do
  s=$(date +%s.%N)
  timeout $T stdbuf -oL -eL /tmp/c_driver_loop > /tmp/c_driver_loop.out 2>&1
  rc=$?
  e=$(date +%s.%N)
  printf "run %2d: rc %d in %.1f s\n" $i $rc $(echo "$e - $s" | bc)
  if [ $rc -ne 0 ]; then echo "---- output of the failing run:"; tail -25 /tmp/c_driver_loop.out; break; fi
done

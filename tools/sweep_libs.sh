#!/bin/bash
# usage: tools/sweep_libs.sh "default w5 w6 ..." "1000000 8000000" [extra bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIBS=${1:-"default"}
SIZES=${2:-"1000000"}
shift 2
for g in $SIZES; do
  for n in $LIBS; do
    lib=$R/raft_rs_amd/libraftgroups_$n.so
    [ "$n" = "default" ] && lib=$R/raft_rs_amd/libraftgroups.so
    [ -f "$lib" ] || { echo "$n: missing $lib"; continue; }
    RG_LIB_PATH=$lib timeout 300 python $R/bench.py --steps 30 --warmup 3 --groups $g --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('lib $n groups $g slots %d wl %d: %.2f Gevals/s  %.1f us/tick  alg %.0f GB/s (%.1f%%)' % (d['config']['peer_slots'], d['config']['workload_id'], d['value']/1e9, r['avg_launch_us'], r['achieved'], 100*r['frac']))
except Exception as e:
    print('lib $n groups $g: FAILED', e)
"
  done
done

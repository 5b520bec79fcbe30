#!/bin/bash
# Run bench.py over the build-time tuning variants (libraftgroups_opt<N>.so) and group counts.
# usage (on the GPU box): tools/sweep_opts.sh "0 1 2 3" "1000000 4000000" [extra bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OPTS=${1:-"0 1 2 3 4 5 6 7"}
SIZES=${2:-"1000000"}
shift 2
for g in $SIZES; do
  for o in $OPTS; do
    lib=$R/raft_rs_amd/libraftgroups_opt$o.so
    [ "$o" = "0" ] && lib=$R/raft_rs_amd/libraftgroups.so
    [ -f "$lib" ] || { echo "opt $o: missing $lib"; continue; }
    out=$(RG_LIB_PATH=$lib timeout 300 python $R/bench.py --steps 30 --warmup 3 --groups $g --no-cpu-baseline "$@" 2>/dev/null)
    echo "$out" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    r=d['roofline']
    print('opt $o groups $g %s: %.2f Gevals/s  %.1f us/tick  alg %.0f GB/s (%.1f%% of 8TB/s)' % (d['config']['kernel_variant'], d['value']/1e9, r['avg_launch_us'], r['achieved'], 100*r['frac']))
except Exception as e:
    print('opt $o groups $g: FAILED', e)
"
  done
done

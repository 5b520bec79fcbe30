#!/bin/bash
# round 4, call x: the default bench line (without the CPU baseline) with the new sub-object between_regimes (2.4 M groups)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04x2
timeout 60 python bench.py --no-cpu-baseline > gpurun_out/r04x2/bench.json 2> gpurun_out/r04x2/err.txt
tail -1 gpurun_out/r04x2/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d['roofline']['by_config'].items(): print(k, v.get('frac'), v.get('us'), v.get('error'))
"

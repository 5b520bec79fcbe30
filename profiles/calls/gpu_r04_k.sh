#!/bin/bash
# round 4, call k: is the slow mode of k_tick_send a matter of where the work-item columns sit relative to the window columns?
# (the RG_SEND_PAD hook -- a padding in front of the work-item columns, read in rg_create -- existed only in the build of this call)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
O=gpurun_out/r04k/pad.txt; : > $O
run() { echo -n "pad=$RG_SEND_PAD $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us' % (d['ms_per_step']*1e3))" >> $O; }
for pad in 0 256 1024 4096 16384 65536 1048576 3145728; do
  export RG_SEND_PAD=$pad
  run --inflights 8 --fused-send
  run --inflights 256 --fused-send
done
cat $O
python - <<'PY'
import raft_rs_amd as rg
for cap in (8, 256):
    e = rg.Engine(1000000, 5, max_inflight=cap)
    print(cap, hex(e.L.rg_column_ptr(e.h, 0) or 0))
PY

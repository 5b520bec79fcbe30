#!/bin/bash
# The round's evidence, on ONE box: rocprofv3 --kernel-trace --stats of every bench configuration, FETCH_SIZE / WRITE_SIZE PMC
# passes (tools/pmc_traffic.sh: separate kernel-trace-only runs, gfx950 correction) for every key of profiles/traffic.json, the
# default bench line. usage (local): /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/gpu_evidence.sh r06'
# then: python tools/merge_traffic.py gpurun_out r06 <commit> tools/gpu_evidence.sh ; cp gpurun_out/<tag>/*_kernel_stats.csv profiles/
cd $GRAFT_REPO_ROOT; TAG=${1:-r06}; export TMPDIR=/tmp
B="--warmup 3 --repeats 1 --no-cpu-baseline --no-extras"
bash tools/gpu_steps.sh $TAG "headline:$TAG:50" \
  "prof:${TAG}_c3:--steps,50,--warmup,5,--repeats,1,--no-cpu-baseline,--no-extras,--workload,3" \
  "prof:${TAG}_c4:--steps,50,--warmup,5,--repeats,1,--no-cpu-baseline,--no-extras,--slots,7" \
  "prof:${TAG}_c5:--steps,50,--warmup,5,--repeats,1,--no-cpu-baseline,--no-extras,--workload,5" \
  "prof:${TAG}_c5_8m:--steps,10,--warmup,3,--repeats,1,--no-cpu-baseline,--no-extras,--workload,5,--groups,8000000" \
  "prof:${TAG}_c5_placed:--steps,30,--warmup,3,--repeats,1,--side,tick,--workload,5,--slots,7,--sorted,--place-after-load" \
  "prof:${TAG}_8m:--steps,12,--warmup,3,--repeats,1,--no-cpu-baseline,--no-extras,--groups,8000000" \
  "prof:${TAG}_2m4:--steps,20,--warmup,3,--repeats,1,--no-cpu-baseline,--no-extras,--groups,2400000" \
  "prof:${TAG}_send2:--steps,50,--warmup,5,--repeats,1,--no-cpu-baseline,--no-extras,--inflights,256" \
  "prof:${TAG}_tick_send:--steps,50,--warmup,5,--repeats,1,--no-cpu-baseline,--no-extras,--inflights,256,--fused-send" \
  "prof:${TAG}_recompute:--steps,50,--warmup,5,--repeats,1,--side,recompute" \
  "prof:${TAG}_gc:--steps,30,--warmup,3,--repeats,1,--side,tick,--group-commit"
pm() { key=$1; steps=$2; shift 2; bash tools/pmc_traffic.sh "$key" $steps "$@" > gpurun_out/$TAG/pmc_$(echo $key | tr ':' '_').txt 2>&1; echo "pmc $key: $(python -c "import json;print(json.load(open('gpurun_out/traffic_$(echo $key | tr ':' '_').json'))['bytes'])" 2>/dev/null)"; }
pm 2:1000000:5 30
pm 3:1000000:5 30 --workload 3
pm 2:1000000:7 30 --slots 7
pm 5:1000000:7:sorted 30 --workload 5
pm 5:1000000:7 30 --workload 5 --size-class-engines
pm 5:1000000:7:one-engine 30 --workload 5 --one-engine
pm 5:8000000:7:sorted 10 --workload 5 --groups 8000000
pm 2:8000000:5 12 --groups 8000000
pm 2:2400000:5 20 --groups 2400000
pm recompute:1000000:5 30 --side recompute
pm recompute:8000000:5 20 --side recompute --groups 8000000
pm 2:1000000:5:inflights 30 --inflights 256
pm 2:1000000:5:inflights:fused-send 30 --inflights 256 --fused-send
pm 2:1000000:5:gc 30 --side tick --group-commit
bash tools/gpu_bench_default.sh $TAG 2>&1 | tail -25
cp gpurun_out/$TAG/bench_default.json gpurun_out/$TAG/${TAG}_bench_n1.json

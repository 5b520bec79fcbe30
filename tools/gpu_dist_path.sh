#!/bin/bash
# The N > 1 code path of bench.py on a ONE-GPU box (no scaling figure: the ranks share the GPU): RCCL at world size 1, then 2 and 8
# ranks started by bench.py itself (`--gpus N` self-launch) with the gloo transport callback behind rg_publish_commit.
# usage: tools/gpu_dist_path.sh <label>   -> gpurun_out/<label>/dist_path.jsonl
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-dist_path}; mkdir -p $O
for cfg in "" "--slots 7"; do
  BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/dist_path.jsonl 2>> $O/dist_path.err
done
BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --groups 500000 --no-cpu-baseline >> $O/dist_path.jsonl 2>> $O/dist_path.err
BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 3 --groups 125000 --slots 7 --no-cpu-baseline >> $O/dist_path.jsonl 2>> $O/dist_path.err
python - "$O" <<'PY'
import json, sys
for l in open(sys.argv[1] + "/dist_path.jsonl"):
    try:
        d = json.loads(l)
        print(d["n_gpus"], "ranks", "%.2f G evals/s" % (d["value"] / 1e9), "%.1f us/step" % (d["ms_per_step"] * 1e3), "|", d["config"].get("sharding", "")[:170])
    except Exception as e:
        print("??", e, l[:120])
PY
tail -3 $O/dist_path.err

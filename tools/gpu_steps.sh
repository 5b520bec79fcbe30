#!/bin/bash
# The box side of tools/gpu_call.sh: gpu_steps.sh <label> <step> ... (see there). Every step is bounded by its own timeout.
label=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 9
export TMPDIR=/tmp
O=gpurun_out/$label; mkdir -p "$O"
sp() { local x="${1//,/ }"; echo "${x//+/,}"; } # (commas separate words; a + stands for a literal comma inside one)
for step in "$@"; do
  IFS=: read -r kind a b c <<< "$step"
  echo "=== $step" | tee -a "$O/steps.txt"
  case $kind in
  tests)
    files=$(sp "${b:-tests}")
    if [ -n "$a" ] && [ "$a" != "-" ]; then timeout 1500 python -m pytest $files -m gpu -x -q -k "$(sp "$a")" 2>&1 | grep -v "^E    .*match\[" | tail -15 > "$O/tests_${c:-0}.txt"
    else timeout 1500 python -m pytest $files -m gpu -x -q 2>&1 | grep -v "^E    .*match\[" | tail -15 > "$O/tests_${c:-0}.txt"; fi
    tail -4 "$O/tests_${c:-0}.txt" ;;
  bench)
    timeout 900 python bench.py $(sp "$a") 2> "$O/bench_err.txt" | tail -1 >> "$O/bench.jsonl"
    tail -1 "$O/bench.jsonl" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
    print('%s | %.2f G/s  %.1f us  frac %.3f' % (d.get('config',{}).get('workload', d.get('workload','?'))[:70], d['value']/1e9, d.get('ms_per_step', d.get('us_per_step',0)/1e3)*1e3, r.get('frac',0)))
except Exception as e: print('??', e)"
    tail -2 "$O/bench_err.txt" ;;
  prof)
    rm -rf /tmp/prof_$a; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$a -o $a --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" $(sp "$b") > "$GRAFT_REPO_ROOT/$O/prof_$a.log" 2>&1 )
    f=$(find /tmp/prof_$a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/${a}_kernel_stats.csv" && head -6 "$O/${a}_kernel_stats.csv" | cut -c1-200 ;;
  headline) # headline:<tag>:<steps> -- the rocprofv3 stats + FETCH_SIZE + WRITE_SIZE passes of the default configuration, condensed
    CMD="python $GRAFT_REPO_ROOT/bench.py --steps ${b:-50} --warmup 5 --repeats 1 --no-cpu-baseline --no-extras"
    P=/tmp/headline_$a; rm -rf $P; mkdir -p $P
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o s -- $CMD > "$GRAFT_REPO_ROOT/$O/headline_bench.json" 2> "$GRAFT_REPO_ROOT/$O/headline_stats.err"
      timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -o f -- $CMD > /dev/null 2> "$GRAFT_REPO_ROOT/$O/headline_fetch.err"
      timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write -o w -- $CMD > /dev/null 2> "$GRAFT_REPO_ROOT/$O/headline_write.err" )
    python tools/summarize_prof.py --tag "$a" --stats $P/stats --fetch $P/fetch --write $P/write --last "${b:-50}" --out "$O" \
      --note "python bench.py --steps ${b:-50} --warmup 5 --repeats 1 --no-cpu-baseline --no-extras (1 M groups x 5 peers, config 2), csrc $(python -c 'import bench; print(bench.csrc_sha16())')" > /dev/null
    f=$(find $P/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/${a}_kernel_stats.csv" && head -4 "$O/${a}_kernel_stats.csv" | cut -c1-160 ;;
  pmc)
    bash tools/pmc_traffic.sh "${a//+/:}" "$b" $(sp "$c") > "$O/pmc_${a//+/_}.txt" 2>&1; tail -3 "$O/pmc_${a//+/_}.txt" ;;
  sweep) # sweep:<libs, + between them>:<name>:<bench args, commas for spaces>  -> sweep.txt (tools/sweep_libs.py)
    timeout 1500 python tools/sweep_libs.py --libs "${a//+/,}" --configs "$b:$(sp "$c")" 2>&1 | tee -a "$O/sweep.txt" ;;
  py)
    timeout 1200 python "$a" $(sp "$b") > "$O/$(basename "$a" .py)${c:+_$c}.txt" 2>&1; tail -40 "$O/$(basename "$a" .py)${c:+_$c}.txt" ;;
  sh)
    timeout 1200 bash -c "$(sp "$a")" 2>&1 | tail -40 ;;
  esac
done

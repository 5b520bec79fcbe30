#!/bin/bash
# tools/gpu_call.sh <label> <timeout-s> <step> [<step> ...] -- ONE parameterised gpurun call (round 5 on; rounds 2-4 kept one
# script per call under profiles/calls/, most of them differing in two lines).
#
# Runs locally: it sends the steps to a fresh MI355X box through gpurun, appends the invocation and its exit code to
# profiles/calls/r06_invocations.log, and leaves whatever the steps wrote under gpurun_out/<label>/ on this side.
# A step is `name[:arg[:arg...]]` (tools/gpu_steps.sh has the bodies):
#   tests:<pytest -k expr or ->:<files...>   pytest -m gpu over the given test files (default: all), tail to tests.txt
#   bench:<args with , for spaces>           python bench.py <args> -> last JSON line appended to bench.jsonl
#   headline:<tag>:<steps>                   rocprofv3 stats + FETCH_SIZE + WRITE_SIZE passes of the default configuration ->
#                                            <tag>_rocprof_summary.{md,json}, <tag>_kernel_stats.csv (tools/summarize_prof.py)
#   prof:<name>:<bench args>                 rocprofv3 --kernel-trace --stats of bench.py <args> -> <name>_kernel_stats.csv
#   pmc:<traffic.json key, + for :>:<steps>:<bench args>   FETCH_SIZE / WRITE_SIZE PMC passes (tools/pmc_traffic.sh ->
#                                            gpurun_out/traffic_<key>.json; fold them in with tools/merge_traffic.py)
#   sweep:<libs, + between>:<name>:<bench args>   tools/sweep_libs.py: experiment builds x one bench configuration -> sweep.txt
#   py:<script.py>:<args>                    python <script> <args> > <script>.txt
#   sh:<command with , for spaces>           anything else
set -u
label=$1; to=$2; shift 2
steps="$*"
cd "$(dirname "$0")/.."
cmd="bash tools/gpu_steps.sh $label $steps"
start=$(date -u +%Y-%m-%dT%H:%M:%SZ)
/usr/local/graft/bin/gpurun --timeout "$to" -- "$cmd"
rc=$?
echo "$start head=$(git rev-parse --short HEAD)$(git diff --quiet || echo +dirty) rc=$rc timeout=$to label=$label steps: $steps" >> profiles/calls/r06_invocations.log
exit $rc

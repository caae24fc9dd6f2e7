#!/bin/bash
# Runs ON THE GPU BOX: the bench lines of every config into gpurun_out/<tag>/ (stderr progress beside them).  usage: tools/final_bench.sh <tag> [configs...]
set -o pipefail
TAG=${1:-r03_final}; shift
CFGS=${@:-"c2 c2r c3 c5 shipped c1"}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$TAG
for c in $CFGS; do
  case $c in
    c2r) ARGS="--config c2 --ragged"; OUT=c2_ragged ;;
    c3|c5) ARGS="--config $c --steps 20 --warmup 5 --cpu-sample 2"; OUT=$c ;;
    shipped) ARGS="--config shipped --steps 10 --warmup 3 --cpu-sample 2"; OUT=shipped ;;
    *) ARGS="--config $c"; OUT=$c ;;
  esac
  echo "== bench.py $ARGS" | tee -a gpurun_out/$TAG/progress.log
  python3 bench.py $ARGS > gpurun_out/$TAG/bench_$OUT.log 2> gpurun_out/$TAG/bench_$OUT.err; echo "rc=$?" | tee -a gpurun_out/$TAG/progress.log
  tail -c 300 gpurun_out/$TAG/bench_$OUT.err | tail -2
done

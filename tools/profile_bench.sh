#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 passes over the default bench command.  usage: tools/profile_bench.sh <config> <tag>
#   1. --kernel-trace --stats            -> per-kernel time
#   2. --pmc FETCH_SIZE ; 3. --pmc WRITE_SIZE (separate passes)   -> HBM-side bytes per launch
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...          -> matrix-core utilisation per kernel
# The program itself follows `--` (python3 bench.py ...): no env / bash -c hop under the profiler.
set -o pipefail
CFG=${1:-c2}; TAG=${2:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_prof_${CFG}
mkdir -p $OUT
ARGS="bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 $ARGS > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 $ARGS > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/busy --output-format csv -- python3 $ARGS > $OUT/busy.log 2>&1 || { tail -5 $OUT/busy.log; exit 1; }
ls $OUT/*/*/ | head -40

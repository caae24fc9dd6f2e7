#!/bin/bash
# Runs ON THE GPU BOX: paired A/B of library variants on one bench config, interleaved rounds.  usage: tools/ab_bench.sh <config> <rounds> <variant...>   ("main" = the in-tree build)
CFG=$1; R=$2; shift 2
cd "$GRAFT_REPO_ROOT"
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = main ]; then L=""; else L="RNNT_HIP_LIB=$PWD/rnntransducer_amd/csrc/librnnt_hip_$v.so"; fi
    env $L python3 bench.py --config $CFG --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']
print('$v', j['ms_per_step'], 'fwd', k['lstm_fwd_kernel']['ms_per_step'], 'bwd', k['lstm_bwd_kernel']['ms_per_step'], 'hp', k.get('gemm_hp_kernel',{}).get('ms_per_step'))"
  done
done

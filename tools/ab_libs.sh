#!/bin/bash
# Runs ON THE GPU BOX: paired bench A/B of the main library against variant builds, no tests / probes.
# usage: tools/ab_libs.sh "NAME1 NAME2" [config] [rounds]   -> gpurun_out/ab_libs.txt
NAMES=$1; CFG=${2:-c2}; ROUNDS=${3:-2}
OUT=gpurun_out/ab_libs.txt
mkdir -p gpurun_out; : > $OUT
for r in $(seq $ROUNDS); do
  for w in main $NAMES; do
    if [ $w = main ]; then L=; else L=$PWD/rnntransducer_amd/csrc/librnnt_hip_$w.so; fi
    RNNT_HIP_LIB=$L timeout -k 10 300 python bench.py --config $CFG --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - $w >> $OUT <<'P'
import json,sys
d=json.loads([l for l in open('gpurun_out/ab_tmp.log') if l.startswith('{')][-1])
k=d['kernels']
print(f"{sys.argv[1]:8s} ms/step {d['ms_per_step']:.3f}  fwd {k['lstm_fwd_kernel']['ms_per_step']:.3f} bwd {k['lstm_bwd_kernel']['ms_per_step']:.3f} hp {k.get('gemm_hp_kernel',{}).get('ms_per_step',0):.3f} loss {d['last_loss']}")
P
  done
done
cat $OUT

#!/bin/bash
# Runs ON THE GPU BOX: paired A/B of environment switches on the bench.  usage: tools/ab_env.sh "VAR1=1 VAR2=1 ..." [config] [rounds]
# (each word is one variant; "none" = defaults)  -> gpurun_out/ab_env.txt
VARS=$1; CFG=${2:-c2}; ROUNDS=${3:-2}
OUT=gpurun_out/ab_env.txt
mkdir -p gpurun_out; : > $OUT
for r in $(seq $ROUNDS); do
  for v in none $VARS; do
    if [ $v = none ]; then E=; else E=${v//+/ }; fi   # "A=1+B=2": several variables in one variant
    env $E timeout -k 10 300 python bench.py --config $CFG --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - "$v" >> $OUT <<'P'
import json,sys
d=json.loads([l for l in open('gpurun_out/ab_tmp.log') if l.startswith('{')][-1])
k=d['kernels']
print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.3f}  fwd {k['lstm_fwd_kernel']['ms_per_step']:.3f} bwd {k['lstm_bwd_kernel']['ms_per_step']:.3f} hp {k.get('gemm_hp_kernel',{}).get('ms_per_step',0):.3f} loss {d['last_loss']}")
P
  done
done
cat $OUT

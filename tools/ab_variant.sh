#!/bin/bash
# Runs ON THE GPU BOX: paired A/B of the main library against variant builds (csrc/build.py --variant NAME).
# usage: tools/ab_variant.sh "NAME1 NAME2" [config] [rounds] [probe=1]   -> gpurun_out/ab_<NAME1>.txt
NAMES=$1; CFG=${2:-c2}; ROUNDS=${3:-2}; PROBE=${4:-1}
OUT=gpurun_out/ab_$(echo $NAMES | tr ' ' '_').txt
mkdir -p gpurun_out
: > $OUT
lib() { if [ $1 = main ]; then echo; else echo $PWD/rnntransducer_amd/csrc/librnnt_hip_$1.so; fi; }
for n in $NAMES; do
  echo "== tests $n" >> $OUT
  RNNT_HIP_LIB=$(lib $n) timeout -k 10 600 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_model.py -x -q 2>&1 | tail -2 >> $OUT || { cat $OUT; exit 1; }
done
if [ $PROBE = 1 ]; then
  for n in main $NAMES; do
    echo "== phase probe $n" >> $OUT; RNNT_HIP_LIB=$(lib $n) timeout -k 10 300 python tools/lstm_phase_probe.py 2>&1 | grep -v "amdgpu.ids\|xcd-local" >> $OUT || exit 1
  done
fi
for r in $(seq $ROUNDS); do
  for w in main $NAMES; do
    RNNT_HIP_LIB=$(lib $w) timeout -k 10 300 python bench.py --config $CFG --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; exit 1; }
    python - $w >> $OUT <<'P'
import json,sys
d=json.loads([l for l in open('gpurun_out/ab_tmp.log') if l.startswith('{')][-1])
k=d['kernels']
print(f"{sys.argv[1]:8s} ms/step {d['ms_per_step']:.3f}  fwd {k['lstm_fwd_kernel']['ms_per_step']:.3f} bwd {k['lstm_bwd_kernel']['ms_per_step']:.3f} hp {k.get('gemm_hp_kernel',{}).get('ms_per_step',0):.3f} loss {d['last_loss']}")
P
  done
done
cat $OUT

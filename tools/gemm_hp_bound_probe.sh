#!/bin/bash
# Runs ON THE GPU BOX: which bound does gemm_hp_kernel sit at?  Same loop with the operand fetch (LDS-DMA) removed, and with the
# MFMAs + LDS reads removed (variant libraries built beforehand in the container:
#   python -m rnntransducer_amd.csrc.build --variant nodma --only=gemm_hp.hip -DHP_DBG_NO_DMA=1
#   python -m rnntransducer_amd.csrc.build --variant nomfma --only=gemm_hp.hip -DHP_DBG_NO_MFMA=1 )
cd "$GRAFT_REPO_ROOT"
python3 tools/gemm_hp_time.py "full kernel"
for v in "$@"; do RNNT_HIP_LIB=$PWD/rnntransducer_amd/csrc/librnnt_hip_$v.so python3 tools/gemm_hp_time.py "$v"; done

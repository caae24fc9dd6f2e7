"""One hp GEMM shape launched a few times, for rocprofv3 --pmc passes (put python3 directly after `--`):
   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d out -- python3 tools/gemm_hp_pmc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm_hp, hp_split
M, N, K = 32000, 4096, 1024
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g)
w = torch.randn(N, K, device="cuda", generator=g) * 0.03
xh, wh = hp_split(x), hp_split(w)
out = torch.empty(M, N, device="cuda")
for _ in range(5):
    gemm_hp(xh, wh, out)
torch.cuda.synchronize()

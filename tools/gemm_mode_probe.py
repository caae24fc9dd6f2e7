"""Runs the three big config-2 GEMM forms (NT input projection, NN dX, TN dW) a few times in the arithmetic mode / tile
variant given by the environment (RNNT_GEMM_MODE, RNNT_GEMM_BK32, RNNT_GEMM_BN128); used under rocprofv3 --pmc.
   python tools/gemm_mode_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda"; M, N, K = 32000, 4096, 1024
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; dg = torch.randn(M, N, device=dev)
o1 = torch.empty(M, N, device=dev); o2 = torch.empty(M, K, device=dev); o3 = torch.empty(N, K, device=dev)
cases = {"NT": lambda: gemm(M, N, K, x, w, o1), "NN": lambda: gemm(M, K, N, dg, w, o2, b_sn=1, b_sk=K),
         "TN": lambda: gemm(N, K, M, dg, x, o3, a_mc=True, a_sk=N, b_sn=1, b_sk=K, split_k=True)}
for name, fn in cases.items():
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{name} {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.1f} TF-eq")

"""A/B of the two gemm_hp kernels (default 256x256 / 2 LDS stages vs RNNT_GEMM_HP_3STAGE=1: 256x128 / 3-stage ring) in ONE process,
interleaved rounds, median reported.  (The same harness measured, on the default kernel: staggered LDS-DMA issue between SIMD partner
waves -5 %, s_setprio around the MFMA clusters -1 %.)   python tools/gemm_hp_ab.py [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm_hp, hp_split
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [(32000, 4096, 1024), (32000, 1024, 4096), (4096, 1024, 32000)]
variants = [("256x256 2-stage (default)", {}), ("256x128 3-stage", {"RNNT_GEMM_HP_3STAGE": "1"})]
for M, N, K in shapes:
    a = hp_split(torch.randn(M, K, device=dev, generator=g))
    b = hp_split(torch.randn(N, K, device=dev, generator=g) * 0.05)
    out = torch.empty(M, N, device=dev)
    times = {n: [] for n, _ in variants}
    for r in range(rounds + 1):
        for name, env in variants:
            os.environ.update(env)
            gemm_hp(a, b, out); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gemm_hp(a, b, out)
            e1.record(); torch.cuda.synchronize()
            if r: times[name].append(e0.elapsed_time(e1) / 5)
            for k in env: del os.environ[k]
    print(f"{M}x{N}x{K}: " + "  ".join(f"{n} {statistics.median(t):.3f} ms ({2.0*M*N*K/statistics.median(t)/1e9:.0f} TF)" for n, t in times.items()))

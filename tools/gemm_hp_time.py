"""Times the half-pair GEMM on the three big c2 shapes with the library RNNT_HIP_LIB points at (default: the in-tree build).
   python tools/gemm_hp_time.py [label]      (used by tools/gemm_hp_bound_probe.sh, one process per library variant)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm_hp, hp_split
label = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
out_line = []
for name, (M, N, K) in (("proj", (32000, 4096, 1024)), ("dX", (32000, 1024, 4096)), ("dW", (4096, 1024, 32000))):
    a = hp_split(torch.randn(M, K, device=dev, generator=g))
    b = hp_split(torch.randn(N, K, device=dev, generator=g) * 0.05)
    out = torch.empty(M, N, device=dev)
    ts = []
    for r in range(6):
        gemm_hp(a, b, out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gemm_hp(a, b, out)
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(e0.elapsed_time(e1) / 5)
    t = statistics.median(ts)
    out_line.append(f"{name} {1e3 * t:.0f} us ({2.0 * M * N * K / t / 1e9:.0f} TF-equivalent)")
print(f"{label:>28} | " + " | ".join(out_line), flush=True)

"""Tile walk order of gemm_hp_kernel (RNNT_GEMM_HP_GROUP_M = 0 column-major | band height) on the c2 / c5 LSTM product shapes:
time per launch only (results are bitwise independent of the walk: tests/test_gpu_gemm.py).  One process per setting, the switch is
read once:    for g in 0 4 8 16; do RNNT_GEMM_HP_GROUP_M=$g python tools/gemm_hp_walk_ab.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm_hp, hp_split

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


line = [f"GROUP_M={os.environ.get('RNNT_GEMM_HP_GROUP_M', 'default')}"]
for tag, M, H2, I in (("c2", 32000, 4096, 1024), ("c5", 24000, 5120, 1280)):
    x = torch.randn(M, I, device=dev, generator=g)
    w = torch.randn(H2, I, device=dev, generator=g) * 0.03
    dg = torch.randn(M, H2, device=dev, generator=g)
    xh, wh, dgh = hp_split(x), hp_split(w), hp_split(dg)
    wth, dgt, xt = hp_split(w, transpose=True), hp_split(dg, transpose=True), hp_split(x, transpose=True)
    o1, o2, o3 = torch.empty(M, H2, device=dev), torch.empty(M, I, device=dev), torch.empty(H2, I, device=dev)
    fl = 2.0 * M * H2 * I
    for name, f in (("proj", lambda: gemm_hp(xh, wh, o1)), ("dX", lambda: gemm_hp(dgh, wth, o2)), ("dW", lambda: gemm_hp(dgt, xt, o3))):
        t = timeit(f)
        line.append(f"{tag}.{name} {1e3 * t:.0f} us {fl / t / 1e9:.0f} TF")
    del x, w, dg, xh, wh, dgh, wth, dgt, xt, o1, o2, o3
print(" | ".join(line))

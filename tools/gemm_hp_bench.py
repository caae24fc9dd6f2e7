"""hp-operand GEMM (csrc/gemm_hp.hip: fp32 through 2 fp16 pieces, 3 MFMA products) on the big GEMM shapes of BASELINE config 2:
time of the split passes and of the product, error against an fp64 product of the same fp32 inputs, next to the default
bf16x6 kernel of gemm.hip.
   python tools/gemm_hp_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm, gemm_hp, hp_split

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
M, N, K = 32000, 4096, 1024
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, generator=g)
w = torch.randn(N, K, device=dev, generator=g) * 0.03
dg = torch.randn(M, N, device=dev, generator=g) * torch.rand(M, 1, device=dev, generator=g).pow(8)  # rows over ~6 decades


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def err(out, ref):
    d = out.double() - ref
    rms = ref.pow(2).mean().sqrt()
    return f"max {float(d.abs().max() / rms):.2e} rms {float(d.pow(2).mean().sqrt() / rms):.2e}"


rows = torch.arange(0, M, 125, device=dev)
cols = torch.arange(0, N, 16, device=dev)
print(f"split x ({M}x{K}) normal: {timeit(lambda: hp_split(x)):.3f} ms; transposed: {timeit(lambda: hp_split(x, transpose=True)):.3f} ms")
print(f"split dG ({M}x{N}) normal: {timeit(lambda: hp_split(dg)):.3f} ms; transposed: {timeit(lambda: hp_split(dg, transpose=True)):.3f} ms")
xh, wh, dgh = hp_split(x), hp_split(w), hp_split(dg)
wth = hp_split(w, transpose=True)       # (K rows, contraction N): W^T
dgt, xt = hp_split(dg, transpose=True), hp_split(x, transpose=True)
out_nt = torch.empty(M, N, device=dev)
out_dx = torch.empty(M, K, device=dev)
out_dw = torch.empty(N, K, device=dev)
cases = [
    ("input projection x.W^T (32000x4096x1024)", 2.0 * M * N * K, lambda: gemm_hp(xh, wh, out_nt), lambda: gemm(M, N, K, x, w, out_nt),
     lambda o: err(o[rows], x[rows].double() @ w.double().t()), out_nt),
    ("dX = dG.W (32000x1024x4096)", 2.0 * M * N * K, lambda: gemm_hp(dgh, wth, out_dx), lambda: gemm(M, K, N, dg, w, out_dx, b_sn=1, b_sk=K),
     lambda o: err(o[rows], dg[rows].double() @ w.double()), out_dx),
    ("dW = dG^T.x (4096x1024x32000, split-K)", 2.0 * M * N * K, lambda: gemm_hp(dgt, xt, out_dw),
     lambda: gemm(N, K, M, dg, x, out_dw, a_mc=True, a_sk=N, b_sn=1, b_sk=K, split_k=True),
     lambda o: err(o[cols], dg[:, cols].double().t() @ x.double()), out_dw),
]
for name, fl, f_hp, f_old, check, out in cases:
    t_old = timeit(f_old)
    e_old = check(out)
    t_hp = timeit(f_hp)
    e_hp = check(out)
    print(f"{name:45s} hp {t_hp:7.3f} ms {fl / t_hp / 1e9:6.1f} TF [{e_hp}] | bf16x6 {t_old:7.3f} ms {fl / t_old / 1e9:6.1f} TF [{e_old}]")
os.environ["RNNT_GEMM_MODE"] = "f32"
for name, fl, f_hp, f_old, check, out in cases:
    f_old()
    print(f"{name:45s} f32-MFMA chain [{check(out)}]")

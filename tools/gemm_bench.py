"""Times the dominant GEMM shapes of BASELINE config 2 through the C ABI in each arithmetic mode (RNNT_GEMM_MODE =
bf16x6 [default] | bf16x3 | f32) and measures each mode's error against an fp64 product of the same fp32 inputs.
   python tools/gemm_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
M, N, K = 32000, 4096, 1024
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, generator=g)
w = torch.randn(N, K, device=dev, generator=g) * 0.03
dg = torch.randn(M, N, device=dev, generator=g)
out_nt = torch.empty(M, N, device=dev)
out_nn = torch.empty(M, K, device=dev)
out_tn = torch.empty(N, K, device=dev)
cases = {
    "NT  x.W^T   (M=32000,N=4096,K=1024)  input projection": lambda: gemm(M, N, K, x, w, out_nt),
    "NN  dG.W    (M=32000,N=1024,K=4096)  dX": lambda: gemm(M, K, N, dg, w, out_nn, b_sn=1, b_sk=K),
    "TN  dG^T.x  (M=4096,N=1024,K=32000)  dW_ih (split-K)": lambda: gemm(N, K, M, dg, x, out_tn, a_mc=True, a_sk=N, b_sn=1, b_sk=K, split_k=True),
    "TN  dG^T.h  (M=2048,N=512,K=32000)   dW_hh (split-K)": lambda: gemm(2048, 512, M, dg, x, out_tn, a_mc=True, a_sk=N, b_sn=1, b_sk=K, split_k=True),
}
flops = {0: 2.0 * M * N * K, 1: 2.0 * M * N * K, 2: 2.0 * M * N * K, 3: 2.0 * 2048 * 512 * M}
rows = torch.arange(0, M, 125, device=dev)  # 256 sample rows


def errors():
    """(max, rms) error relative to the rms magnitude of the exact result, per GEMM form"""
    out = []
    ref = x[rows].double() @ w.double().t()
    out.append(("NT", out_nt[rows].double() - ref, ref))
    ref = dg[rows].double() @ w.double()
    out.append(("NN", out_nn[rows].double() - ref, ref))
    cols = torch.arange(0, N, 16, device=dev)
    ref = dg[:, cols].double().t() @ x.double()
    out.append(("TN", out_tn[cols].double() - ref, ref))
    return "  ".join(f"{n}: max {float(d.abs().max() / r.pow(2).mean().sqrt()):.2e} rms {float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()):.2e}"
                     for n, d, r in out)


for mode in ("bf16x6", "bf16x3", "f32"):
    os.environ["RNNT_GEMM_MODE"] = mode
    print(f"--- RNNT_GEMM_MODE={mode}")
    for i, (name, fn) in enumerate(cases.items()):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        print(f"{name:62s} {ms:7.3f} ms  {flops[i] / ms / 1e9:7.1f} TFLOP/s (fp32-equivalent)")
    list(cases.values())[2]()  # out_tn holds the full dW_ih product again
    torch.cuda.synchronize()
    print("error vs fp64 / rms(|exact|):  " + errors())

"""Times every GEMM shape of one BASELINE-config-2 training step through the C ABI; prints per-shape ms and TFLOP/s.
   python tools/gemm_shapes_c2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm
from rnntransducer_amd._lib import GEMM_GELU_A, GEMM_GELU_B, GEMM_MUL_DGELU

dev = "cuda"
T, B, H, O, V, U1 = 1000, 32, 512, 512, 72, 41
M = T * B
def buf(*shape): return torch.randn(*shape, device=dev) * 0.1
big = buf(M, 4096); x1024 = buf(M, 1024); x80 = buf(M, 80); w = buf(4096, 1024); w80 = buf(4096, 80); out4096 = torch.empty(M, 4096, device=dev)
o1024 = torch.empty(M, 1024, device=dev); dw = torch.empty(4096, 1024, device=dev); wo = buf(512, 1024); y512 = buf(M, 512); wfc = buf(72, 1024)
a72 = buf(M, 72); pm = U1 * B; px = buf(pm, 512); pw = buf(2048, 512); pg = buf(pm, 2048)
cases = [
 ("fwd  L0 input proj  NT 32000x4096x80   x1", 1, 2.0*M*4096*80,  lambda: gemm(M, 4096, 80, x80, w80, out4096)),
 ("fwd  L1-3 input proj NT 32000x4096x1024 x3", 3, 2.0*M*4096*1024, lambda: gemm(M, 4096, 1024, x1024, w, out4096)),
 ("fwd  out_proj       NT 32000x512x1024   x1", 1, 2.0*M*512*1024, lambda: gemm(M, 512, 1024, x1024, wo, y512)),
 ("fwd  joint A (gelu) NT 32000x72x512     x1", 1, 2.0*M*72*512,   lambda: gemm(M, 72, 512, y512, wfc, a72, b_sn=1024, b_sk=1, flags=GEMM_GELU_A)),
 ("fwd  pred input proj NT 1312x2048x512   x1", 1, 2.0*pm*2048*512, lambda: gemm(pm, 2048, 512, px, pw, pg)),
 ("bwd  dX L1-3        NN 32000x1024x4096  x3", 3, 2.0*M*4096*1024, lambda: gemm(M, 1024, 4096, big, w, o1024, b_sn=1, b_sk=1024)),
 ("bwd  dW_ih L1-3     TN 4096x1024x32000  x3", 3, 2.0*M*4096*1024, lambda: gemm(4096, 1024, M, big, x1024, dw, a_mc=True, a_sk=4096, b_sn=1, b_sk=1024, split_k=True)),
 ("bwd  dW_ih L0       TN 4096x80x32000    x1", 1, 2.0*M*4096*80,  lambda: gemm(4096, 80, M, big, x80, w80, a_mc=True, a_sk=4096, b_sn=1, b_sk=80, split_k=True)),
 ("bwd  dW_hh          TN 2048x512x31968   x8", 8, 2.0*(M-B)*2048*512, lambda: gemm(2048, 512, M - B, big, x1024, dw, a_mc=True, a_sk=4096, b_sn=1, b_sk=1024, split_k=True)),
 ("bwd  out_proj dX    NN 32000x1024x512   x1", 1, 2.0*M*512*1024, lambda: gemm(M, 1024, 512, y512, wo, o1024, b_sn=1, b_sk=1024)),
 ("bwd  out_proj dW    TN 512x1024x32000   x1", 1, 2.0*M*512*1024, lambda: gemm(512, 1024, M, y512, x1024, wo, a_mc=True, a_sk=512, b_sn=1, b_sk=1024, split_k=True)),
 ("bwd  joint d_enc    NN 32000x512x72     x1", 1, 2.0*M*72*512,   lambda: gemm(M, 512, 72, a72, wfc, y512, b_sn=1, b_sk=1024, aux=y512, flags=GEMM_MUL_DGELU)),
 ("bwd  joint dW_e     TN 72x512x32000     x1", 1, 2.0*M*72*512,   lambda: gemm(72, 512, M, a72, y512, wfc, a_mc=True, a_sk=72, b_sn=1, b_sk=512, c_div=1, c_so=1024, c_si=0, flags=GEMM_GELU_B, split_k=True)),
]
tot_ms = tot_fl = 0.0
for name, n, fl, fn in cases:
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    tot_ms += n * ms; tot_fl += n * fl
    print(f"{name:46s} {ms:7.3f} ms  {fl/ms/1e9:7.1f} TF/s   step share {n*ms:6.2f} ms")
print(f"sum over one step: {tot_ms:.2f} ms, {tot_fl/tot_ms/1e9:.1f} TF/s average")

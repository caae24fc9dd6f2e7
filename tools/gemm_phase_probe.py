"""In-kernel phase timers of the split-bf16 GEMM (debug instantiation, RNNT_GEMM_DBG=1): cycles a wave spends per K-tile in
barrier 1 | vmcnt wait | split + LDS store | barrier 2 | global-load issue | LDS operand reads + MFMA.
   RNNT_GEMM_DBG=1 python tools/gemm_phase_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd import _lib
os.environ["RNNT_GEMM_DBG"] = "1"
dev = "cuda"
def run(name, M, N, K, A, B, **kw):
    out = torch.empty(M, N, device=dev)
    ws = torch.zeros(32 * 4 * 8, dtype=torch.int64, device=dev)
    d = _lib.GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.A, d.a_div, d.a_so, d.a_si, d.a_sk, d.a_mc = A.data_ptr(), 1 << 40, 0, kw.get("a_si", K), kw.get("a_sk", 1), kw.get("a_mc", 0)
    d.B, d.b_sn, d.b_sk = B.data_ptr(), kw.get("b_sn", K), kw.get("b_sk", 1)
    d.C, d.c_div, d.c_so, d.c_si = out.data_ptr(), 1 << 40, 0, N
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 8
    for _ in range(3):
        _lib.check(_lib.lib().rnnt_hip_gemm_f32(C.byref(d), torch.cuda.current_stream().cuda_stream), "gemm")
    torch.cuda.synchronize()
    t = ws.view(32, 4, 8).cpu().double()
    t = t[t[:, 0, 6] > 0]  # only the sampled workgroups that exist in this grid
    nk = t[0, 0, 6].item()
    per = t[:, :, :6].mean(dim=(0, 1)) / nk
    names = ["barrier1", "vmcnt", "split+store", "barrier2", "load issue", "ds_read+mfma"]
    print(f"{name}: K-tiles {int(nk)}  cycles per K-tile per wave: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, per.tolist())) + f"  total {per.sum().item():.0f}")
M, N, K = 32000, 4096, 1024
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; dg = torch.randn(M, N, device=dev)
run("NT 32000x4096x1024", M, N, K, x, w)
run("NN 32000x1024x4096", M, K, N, dg, w, b_sn=1, b_sk=K)
run("TN 4096x1024x32000 (no split-K)", N, K, M, dg, x, a_mc=1, a_sk=N, b_sn=1, b_sk=K)

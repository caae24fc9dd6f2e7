"""Diagnostic: every gemm.hip product of one c2 training step (shape, flags, time) — which products are still off the half-pair path."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rnntransducer_amd import ops
from rnntransducer_amd.data import synthetic_batch

cfg = bench.CONFIGS[os.environ.get("CFG", "c2")]
B, T, U, V = cfg[:4]
model, tn, pn = bench.build_model(cfg, 0.2, 100)
model = model.cuda().train()
batch = synthetic_batch(B, T, U, V, ragged=False, seed=1234, device=torch.device("cuda"))
conf = model.configure_optimizers()
opt = conf["optimizer"]
rec = []
orig = ops.gemm
def traced(M, N, K, A, Bm, Cout, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(M, N, K, A, Bm, Cout, **kw); e1.record()
    rec.append((M, N, K, kw.get("flags", 0), bool(kw.get("a_mc")), bool(kw.get("split_k")), e0, e1))
def step():
    opt.zero_grad(); loss = model.training_step(batch, 0)["loss"]; loss.backward(); opt.step()
for _ in range(3): step()
ops.gemm = traced
for m in list(sys.modules.values()):
    if m and getattr(m, "gemm", None) is orig: m.gemm = traced
rec.clear(); step(); torch.cuda.synchronize()
tot = 0.0
for M, N, K, fl, mc, sk, e0, e1 in rec:
    ms = e0.elapsed_time(e1); tot += ms
    print(f"M={M:6d} N={N:5d} K={K:6d} flags={fl:2d} a_mc={int(mc)} split_k={int(sk)}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF")
print(f"{len(rec)} python-level gemm calls, {tot:.3f} ms (LSTM-internal gemm.hip launches are not listed)")

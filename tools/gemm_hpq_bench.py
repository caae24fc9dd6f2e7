"""Weight-gradient products of ONE BASELINE-config-2 encoder layer (dW_ih 4096x1024, dW_hh 2 x 2048x512, contraction T*B = 32000):
three launches of gemm_hp against the grouped queue-driven launch on all XCDs / on XCDs 4-7 only, alone on the device and while a
spinning kernel (a stand-in for a persistent recurrence: 128 workgroups that only poll a word) holds XCDs 0-3.
   python tools/gemm_hpq_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.ops import gemm_hp, gemm_hp_grouped, hp_split

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
M, N4, I, H = 32000, 4096, 1024, 512
g = torch.Generator(device=dev).manual_seed(0)
dg = torch.randn(M, N4, device=dev, generator=g) * 0.01
x = torch.randn(M, I, device=dev, generator=g)
y = torch.randn(M, 2 * H, device=dev, generator=g)
dgt = hp_split(dg, transpose=True)
dgt0, dgt1 = hp_split(dg[:, :4 * H].contiguous(), transpose=True), hp_split(dg[:, 4 * H:].contiguous(), transpose=True)
xt = hp_split(x, transpose=True)
yt0, yt1 = hp_split(y[:, :H].contiguous(), transpose=True, shift=-32), hp_split(y[:, H:].contiguous(), transpose=True, shift=32)
pairs = [(dgt, xt), (dgt0, yt0), (dgt1, yt1)]
outs = [torch.empty(a.rows, b.rows, device=dev) for a, b in pairs]
flops = sum(2.0 * a.rows * b.rows * a.K for a, b in pairs)


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


def separate():
    for (a, b), o in zip(pairs, outs):
        gemm_hp(a, b, o)


ref = [o.clone() for o in (separate() or outs)]
for name, fn in (("3 launches of gemm_hp", separate),
                 ("grouped, all XCDs", lambda: gemm_hp_grouped(pairs, outs)),
                 ("grouped, XCDs 4-7 only", lambda: gemm_hp_grouped(pairs, outs, xcd_skip=0x0F)),
                 ("grouped, XCDs 2-7 only", lambda: gemm_hp_grouped(pairs, outs, xcd_skip=0x03))):
    t = timeit(fn)
    fn()
    torch.cuda.synchronize()
    dev_max = max(float((o - r).abs().max() / r.abs().max()) for o, r in zip(outs, ref))
    print(f"{name:28s} {t:7.3f} ms  {3 * flops / t / 1e9:8.1f} TF/s f16-equivalent   max |diff| vs separate / max|ref| = {dev_max:.1e}")

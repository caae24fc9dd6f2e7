"""What stretches a persistent recurrence when other work runs beside it on the OTHER XCDs?  One config-2 bi-LSTM layer forward
(XCD-local v5 recurrence on XCDs 0-3) alone, beside matrix-core-only work on XCDs 4-7, beside HBM streaming on XCDs 4-7, and
beside the real grouped weight-gradient GEMM.   python tools/interfere_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnntransducer_amd.networks.rnn import HipLSTM
from rnntransducer_amd.ops import gemm_hp_grouped, hp_split

here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libinterfere.so"))
T, B, I, H = 1000, 32, 1024, 512
torch.manual_seed(0)
lstm = HipLSTM(I, H, 1, bidirectional=True).cuda()
x = torch.randn(T, B, I, device="cuda")
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
out = torch.zeros(16, device="cuda")
big = torch.randn(1 << 28, device="cuda")   # 1 GiB
side = torch.cuda.Stream()
M = T * B
dg = torch.randn(M, 4096, device="cuda") * 0.01
pairs = [(hp_split(dg, transpose=True), hp_split(torch.randn(M, 1024, device="cuda"), transpose=True))]
outs = [torch.empty(4096, 1024, device="cuda")]


x.requires_grad_(True)
dy = torch.randn(T, B, 2 * H, device="cuda")
state = {}


def rec():
    # backward of the layer: the reverse-time recurrence comes FIRST (2.2 ms), then its GEMMs — side work of <= 2 ms launched at
    # the same moment runs beside the recurrence only
    torch.autograd.grad(state.pop("y"), [x] + list(lstm.parameters()), dy)


def run(side_fn, reps=5):
    times, side_times = [], []
    for _ in range(reps + 1):
        state["y"] = lstm(x, lens)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rec()
        b.record()
        if side_fn is not None:   # submitted after the whole backward was enqueued: the recurrence is already resident when it starts
            with torch.cuda.stream(side):
                c.record()
                side_fn(side.cuda_stream)
                d.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
        side_times.append(c.elapsed_time(d) if side_fn is not None else 0.0)
    return sorted(times[1:])[len(times[1:]) // 2], sorted(side_times[1:])[len(side_times[1:]) // 2]


def grouped(_s):
    gemm_hp_grouped(pairs, outs, xcd_skip=0x0F)


cases = [("layer backward alone (v5 recurrence, then GEMMs)", None),
         ("+ MFMA-only work on XCDs 4-7", lambda s: L.interfere_mfma_spin(0x0F, 15000, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(s))),
         ("+ MFMA-only work on ALL XCDs' free slots", lambda s: L.interfere_mfma_spin(0x00, 15000, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(s))),
         ("+ HBM streaming on XCDs 4-7 (8 GiB)", lambda s: L.interfere_mem_stream(0x0F, ctypes.c_void_p(big.data_ptr()), ctypes.c_long(big.numel() * 4), 8, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(s))),
         ("+ grouped dW_ih GEMM on XCDs 4-7", grouped)]
for name, fn in cases:
    t, ts = run(fn)
    print(f"{name:50s} recurrence call {t:7.3f} ms   side work {ts:7.3f} ms")

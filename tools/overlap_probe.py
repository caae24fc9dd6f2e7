"""Can weight-gradient GEMMs hide under a persistent recurrence?  Times one config-2 bi-LSTM layer forward (3 ms persistent
kernel, 1 workgroup per CU, 16 KB LDS, 256 VGPRs) and three dW-shaped split-bf16 GEMMs serially on one stream, then the
same work with the GEMMs on a second stream.   python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RNNT_GEMM_NO256"] = "1"  # the 8-wave / 147 KB tile cannot share a CU with the recurrence; the 4-wave one can
import torch
from rnntransducer_amd.ops import gemm
from rnntransducer_amd.networks.rnn import HipLSTM

T, B, I, H = 1000, 32, 1024, 512
torch.manual_seed(0)
lstm = HipLSTM(I, H, 1, bidirectional=True).cuda()
x = torch.randn(T, B, I, device="cuda")
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
M = T * B
dg = torch.randn(M, 4096, device="cuda"); xx = torch.randn(M, 1024, device="cuda"); dw = torch.empty(4096, 1024, device="cuda")
side = torch.cuda.Stream()

def rec():
    with torch.no_grad():
        return lstm(x, lens)
def gemms(n=2):
    for _ in range(n):
        gemm(4096, 1024, M, dg, xx, dw, a_mc=True, a_sk=4096, b_sn=1, b_sk=1024, split_k=True)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def serial():
    rec(); gemms()
def overlapped():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        gemms()
    rec()
    torch.cuda.current_stream().wait_stream(side)

print(f"recurrence + input GEMM alone   {timed(rec):7.3f} ms")
print(f"2 dW GEMMs alone                {timed(gemms):7.3f} ms")
print(f"serial, one stream              {timed(serial):7.3f} ms")
print(f"GEMMs on a second stream        {timed(overlapped):7.3f} ms")

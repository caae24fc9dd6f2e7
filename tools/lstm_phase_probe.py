"""Diagnostic: per-phase cycle shares of one step of the persistent LSTM recurrences (v5 by default; RNNT_LSTM_NO_V5=1: v3/v4) at BASELINE config 2 layer shapes.
   The per-phase stamps are compiled out of the default library (they cost the step loops 3.5-4.5 %): build the variant first, in the container,
   python -m rnntransducer_amd.csrc.build --variant dbg --only=lstm.hip,lstm5.hip -DRNNT_LSTM_DBG_STAMPS=1
   and run this script on the GPU box; it loads librnnt_hip_dbg.so by itself."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RNNT_LSTM_DBG"] = "1"
_dbg_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rnntransducer_amd", "csrc", "librnnt_hip_dbg.so")
if not os.environ.get("RNNT_HIP_LIB"):
    if not os.path.exists(_dbg_lib):
        raise SystemExit("build the stamped variant first: python -m rnntransducer_amd.csrc.build --variant dbg --only=lstm.hip,lstm5.hip -DRNNT_LSTM_DBG_STAMPS=1")
    os.environ["RNNT_HIP_LIB"] = _dbg_lib
import numpy as np
import torch
from rnntransducer_amd import _lib
from rnntransducer_amd.ops import LstmStackFn, _addr, _stream
from rnntransducer_amd.networks.rnn import HipLSTM

T, B, I, H = 1000, int(os.environ.get("PROBE_B", "32")), 1024, int(os.environ.get("PROBE_H", "512"))
torch.manual_seed(0)
lstm = HipLSTM(I, H, 1, bidirectional=True).cuda()
x = torch.randn(T, B, I, device="cuda", requires_grad=True)
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
names = ["prefetch issue", "flag wait", "gather+MFMA", "reduce+cell math", "drain+barrier+flag", "stash stores"]
names5b = ["loop top (dropout mask)", "tagged poll (wait for partial dh)", "partial sums + barrier", "cell math + dG image + barrier",
           "scale + MFMA + publication", "gather issue + prefetch + stash"]
names5 = ["loop top", "tagged poll (wait for operands)", "de-interleave + MFMA", "partial write + barrier", "reduce + cell math + publish",
          "gather issue + stash + prefetch"]

class Hook:
    pass

def read(ws, tag):
    out = np.zeros((256, 8), dtype=np.uint64)
    _lib.check(_lib.lib().rnnt_hip_lstm_debug_read(_addr(ws), T, B, I, H, 2, out.ctypes.data, 256, _stream()), "dbg")
    tot = out[:, :6].sum(1).astype(np.float64)
    print(f"--- {tag}: cycles/step per phase (median over 256 workgroups; min..max), total {np.median(tot) / T:.0f} cyc/step")
    print(f"   xcd-local groups: {int(out[:, 6].sum())}/256 workgroups; XCC ids of blocks 0..15: {out[:16, 7].astype(int).tolist()}")
    v5 = not os.environ.get("RNNT_LSTM_NO_V5")
    for i, n in enumerate((names5 if tag == "forward" else names5b) if v5 else names):
        v = out[:, i].astype(np.float64) / T
        print(f"   {n:22s} {np.median(v):8.0f}   ({v.min():.0f} .. {v.max():.0f})")

for it in range(2):
    y = lstm(x, lens)
    torch.cuda.synchronize()
ws = y.grad_fn.ws
read(ws, "forward")
y.backward(torch.randn_like(y))
torch.cuda.synchronize()
read(ws, "backward")

import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from rnntransducer_amd.networks.rnn import HipLSTM
    T, B, I, H = 30, 4, 80, 128
    torch.manual_seed(0)
    lstm = HipLSTM(I, H, 1, bidirectional=True).cuda()
    x = torch.randn(T, B, I, device="cuda", requires_grad=True)
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    y = lstm(x, lens)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda()
    y.backward(g)
    torch.save({"dx": x.grad.cpu(), **{n: p.grad.cpu() for n, p in lstm.named_parameters()}}, sys.argv[1])
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/new.pt"], env=env)
    env["RNNT_LSTM_BWD5_2B"] = "1"
    subprocess.check_call([sys.executable, __file__, "/tmp/old.pt"], env=env)
    import torch
    a, b = torch.load("/tmp/new.pt"), torch.load("/tmp/old.pt")
    for k in a:
        d = (a[k] - b[k]).abs()
        print(k, tuple(a[k].shape), "max|new-old|", d.max().item(), "max|old|", b[k].abs().max().item())
    d = (a["dx"] - b["dx"]).abs()   # (T,B,I)
    print("dx err per t:", d.amax(dim=(1, 2)).tolist())
    print("dx err per b:", d.amax(dim=(0, 2)).tolist())

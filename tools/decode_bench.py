"""Times JointNet.recognize_greedy (one kernel launch per batch, csrc/decode.hip) at the config-2 layer sizes and, on a
bounded sample, the CPU oracle's host-loop restatement of networks/transducer.py:95-145.

    python tools/decode_bench.py [--batch 32] [--frames 1000] [--reps 5] [--cpu-utts 1]

Weights are random-init scaled so the search emits symbols (random init alone decodes to nothing); data synthetic.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-utts", type=int, default=1)
    ap.add_argument("--blank-bias", type=float, default=0.0, help="added to fc.bias[blank]: fewer emitted symbols per frame")
    a = ap.parse_args()
    from oracle.rnnt_oracle import OracleJointNet
    from rnntransducer_amd import _lib
    from rnntransducer_amd.networks import JointNet
    tn = dict(input_size=80, hidden_size=512, output_size=320, num_layers=3, rnn_type="lstm", dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=72, pad_token_id=0, hidden_size=512, output_size=320, num_layers=1, rnn_type="lstm", dropout=0.0)
    torch.manual_seed(0)
    net = JointNet(dict(tn), dict(pn), 72)
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.mul_(4.0 if n.startswith("fc.") else 2.0)
        net.decoder.embedding.weight[0].zero_()
        net.fc.bias[0] += a.blank_bias
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.cuda().eval()
    audios = torch.randn(a.batch, a.frames, 80)
    lens = [a.frames] * a.batch
    dev_audio = audios.cuda()
    out = net.recognize_greedy(dev_audio, lens, 0, 3)
    torch.cuda.synchronize()
    ntok = [int(x.numel()) for x in (out if isinstance(out, list) else [out[0]])]
    lib = _lib.lib()
    lib.rnnt_hip_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        net.recognize_greedy(dev_audio, lens, 0, 3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    import ctypes as C
    n = len(_lib.KERNEL_KINDS)
    ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
    lib.rnnt_hip_prof_collect(ms, work, cnt, n)
    lib.rnnt_hip_prof_enable(0)
    per_kind = {k: round(ms[i] / a.reps, 3) for i, k in enumerate(_lib.KERNEL_KINDS) if cnt[i]}
    ora = OracleJointNet(dict(tn), dict(pn), 72).eval()
    ora.load_state_dict(sd)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.perf_counter()
    want = ora.recognize_greedy(audios[:a.cpu_utts], lens[:a.cpu_utts], 0, 3)
    cpu_dt = time.perf_counter() - t0
    got = out if isinstance(out, list) else [out[0]]
    agree = sum(int(got[b].tolist() == want[b]) for b in range(a.cpu_utts))
    print(json.dumps({"metric": "greedy decode utterances/sec", "value": round(a.batch / dt, 2), "ms_per_batch": round(dt * 1e3, 2),
                      "batch": a.batch, "frames": a.frames, "blank_bias": a.blank_bias, "tokens_per_utt": sum(ntok) / len(ntok),
                      "kernel_ms_per_batch": per_kind,
                      "cpu_baseline": {"value": round(a.cpu_utts / cpu_dt, 3), "unit": "utterances/sec", "kind": "port",
                                       "sample": f"{a.cpu_utts} utterance(s), torch CPU host loop", "cores": torch.get_num_threads()},
                      "token_agreement_on_cpu_sample": f"{agree}/{a.cpu_utts}", "dtype": "f32", "data": "synthetic"}))


if __name__ == "__main__":
    main()

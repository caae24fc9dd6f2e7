"""Times LogMelFrontend on a config-2-shaped batch of waveforms (B x 10 s @ 16 kHz) and the CPU oracle on a bounded sample.
   python tools/frontend_bench.py [--batch 32] [--seconds 10] [--reps 20]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.frontend_oracle import log_mel
from rnntransducer_amd.frontend import LogMelFrontend

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
L = int(a.seconds * 16000)
wav = torch.randn(a.batch, L)
lens = [L] * a.batch
fe = LogMelFrontend().cuda()
dev = wav.cuda()
feats, nfr = fe(dev, lens)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    fe(dev, lens)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
t0 = time.perf_counter()
want = log_mel(wav[0].numpy())
cpu = time.perf_counter() - t0
err = float((feats[0, :nfr[0]].double().cpu() - torch.from_numpy(want)).abs().max())
algo_bytes = a.batch * (L * 4 + feats.shape[1] * 80 * 4)
print(json.dumps({"metric": "log-mel front-end utterances/sec", "value": round(a.batch / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                  "batch": a.batch, "seconds_per_utt": a.seconds, "frames_per_utt": int(feats.shape[1]),
                  "algorithmic_GB_s": round(algo_bytes / dt / 1e9, 1),
                  "cpu_baseline": {"value": round(1.0 / cpu, 1), "unit": "utterances/sec", "kind": "port", "sample": "1 utterance, torch.stft float64 + numpy", "cores": torch.get_num_threads()},
                  "max_abs_err_vs_oracle": err, "dtype": "f32", "data": "synthetic"}))

#!/usr/bin/env python3
"""Where does the c2 parity leg's gradient deviation come from?  Trains the bench model for N steps, then evaluates loss +
the three probe gradients on 2 utterances with dropout off under several arithmetic switches, each against the float64
oracle; the fp32 CPU oracle (torch fp32 = the reference's own arithmetic) is measured against float64 on the same data too.
    python tools/parity_probe.py [--steps 60]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--config", default="c2")
    a = ap.parse_args()
    from oracle.rnnt_oracle import training_loss
    from rnntransducer_amd.data import synthetic_batch
    cfg = bench.CONFIGS[a.config]
    B, T, U, V = cfg[:4]
    dev = torch.device("cuda", 0)
    model, tn, pn = bench.build_model(cfg, 0.2, max(100, a.steps + 1))
    model = model.to(dev).train()
    batch = synthetic_batch(B, T, U, V, ragged=False, seed=1234, device=dev)
    conf = model.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    for _ in range(a.steps):
        opt.zero_grad()
        model.training_step(batch, 0)["loss"].backward()
        opt.step()
        sched.step()
    torch.cuda.synchronize()
    cores, _ = bench.host_cpu_info()
    torch.set_num_threads(cores)
    nb = 2
    sub = tuple((x[:nb] if isinstance(x, torch.Tensor) else x[:nb]) for x in batch)
    cpu_sub = tuple((x.cpu() if isinstance(x, torch.Tensor) else x) for x in sub)
    o64 = bench._oracle_for(model, tn, pn, V, double=True)
    ref = training_loss(o64, (cpu_sub[0].double(),) + cpu_sub[1:])
    ref.backward()
    g64 = {k: p.grad for k, p in o64.named_parameters()}
    o32 = bench._oracle_for(model, tn, pn, V, double=False)
    l32 = training_loss(o32, cpu_sub)
    l32.backward()
    rows = {}

    def report(tag, loss, grads):
        r = {"loss_rel": abs(float(loss) - float(ref)) / abs(float(ref))}
        worst = ("", 0.0)
        for k, g in grads.items():
            rel = float((g.double().cpu() - g64[k]).abs().max() / g64[k].abs().max().clamp_min(1e-30))
            if k in bench.GRAD_PROBES:
                r[k] = rel
            if rel > worst[1]:
                worst = (k, rel)
        r["worst"] = f"{worst[0]} {worst[1]:.2e}"
        rows[tag] = r
        print(tag, json.dumps(r), flush=True)

    report("torch-CPU fp32 oracle", l32.detach(), {k: p.grad for k, p in o32.named_parameters()})
    model.eval()
    for tag, env in (("HIP default", {}), ("HIP RNNT_LSTM_EXACT_MATH=1", {"RNNT_LSTM_EXACT_MATH": "1"}),
                     ("HIP RNNT_GEMM_MODE=f32", {"RNNT_GEMM_MODE": "f32"}),
                     ("HIP all-f32-MFMA + exact math", {"RNNT_GEMM_MODE": "f32", "RNNT_LSTM_V2": "1", "RNNT_LSTM_EXACT_MATH": "1"})):
        os.environ.update(env)
        for p in model.parameters():
            p.grad.zero_()
        l = model.jointnet.loss(sub[0], sub[2], sub[3], sub[5], sub[6], model.blank_token_id).mean()
        l.backward()
        torch.cuda.synchronize()
        report(tag, l.detach(), {k: p.grad.detach().clone() for k, p in model.jointnet.named_parameters()})
        for k in env:
            del os.environ[k]
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "parity_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

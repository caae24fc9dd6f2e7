#!/usr/bin/env python3
"""bench.py — utterances/sec of the full RNN-T training step on the MI355X HIP hot path.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                (N>1, one rank per GPU, RCCL over xGMI)

A "step" = forward (LSTM encoder + LSTM prediction net) + fused joint/RNN-T loss + backward + flat-gradient
all-reduce + AdamW, on one synthetic batch that is already resident in HBM.  Workload at every N: BASELINE.json
configs[1] per GPU — B=32, T=1000 (10 s @ 80 mel), U=40, V=72, 4x512 bi-LSTM encoder / 1x512 LSTM prediction net,
fp32, inter-layer dropout 0.2.  Weak scaling: per-GPU work fixed; value = total utterances / s.

Rank 0 prints ONE JSON line with the driver's contract plus
  "roofline"     for the kernel that took the most time inside the timed region (live HIP-event timing from the
                 library's opt-in profiler; algorithmic FLOPs/bytes per launch are computed in the launch wrappers),
  "cpu_baseline" the CPU oracle (torch-CPU composite of the reference path + C loss) timed on this box's host cores
                 on a bounded sample, N=1 only,
  "loss_rel_delta" |L_hip - L_oracle| / L_oracle on that same sample (dropout off on both sides).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    # name: (B, T, U, V, enc(H, L), pred(H, L), O)
    "c1": (2, 100, 20, 72, (128, 1), (128, 1), 128),
    "c2": (32, 1000, 40, 72, (512, 4), (512, 1), 512),
    "c3": (8, 2000, 120, 72, (512, 4), (512, 1), 512),
    "c5": (16, 1500, 80, 2048, (640, 6), (640, 1), 640),
    # the reference's SHIPPED config/config.json (8x1024 bi-GRU encoder, 2x1024 LSTM prediction net, O=512), B=16
    "shipped": (16, 1000, 40, 72, (1024, 8, "gru"), (1024, 2, "lstm"), 512),
}
GRAD_PROBES = ("fc.weight", "encoder.rnn.weight_hh_l0", "decoder.embedding.weight")  # SURVEY §8(d) "Loss delta" row
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec


def build_model(cfg, dropout, total_steps):
    from argparse import Namespace

    from rnntransducer_amd import RNNTransducer
    B, T, U, V, enc, pred, O = cfg
    He, Le, Te = (tuple(enc) + ("lstm",))[:3]
    Hp, Lp, Tp = (tuple(pred) + ("lstm",))[:3]
    tn = dict(input_size=80, hidden_size=He, output_size=O, num_layers=Le, rnn_type=Te, dropout=dropout, bidirectional=True)
    pn = dict(embedding_size=V, hidden_size=Hp, output_size=O, num_layers=Lp, rnn_type=Tp, dropout=dropout)
    args = Namespace(learning_rate=1e-3, weight_decay=1e-4, warmup_ratio=0.2, final_div_factor=1e4, total_steps=total_steps,
                     move_metrics_to_cpu=False)
    torch.manual_seed(0)  # same initial weights on every rank (DDP broadcasts rank 0's; same seed is equivalent)
    return RNNTransducer(pn, tn, dict(num_classes=V), args), tn, pn


def host_cpu_info():
    """(usable cores, model string): cores = CPUs this process may run on (affinity mask, further clamped by a cgroup
    cpu.max quota if one is set) — what torch.set_num_threads() gets; the model string is /proc/cpuinfo's."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, model


def progress(msg):
    """Heartbeat on stderr (stdout carries exactly one JSON line): the CPU legs of the larger configs run for minutes."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _oracle_for(model, tn, pn, V, double):
    from oracle.rnnt_oracle import OracleJointNet
    oracle = OracleJointNet(dict(tn, dropout=0.0), dict(pn, pad_token_id=0, dropout=0.0), V)
    if double:
        oracle = oracle.double()
    oracle.load_state_dict({k[len("jointnet."):]: (v.detach().cpu().double() if double else v.detach().cpu())
                            for k, v in model.state_dict().items()})
    return oracle


def cpu_baseline(model, tn, pn, V, batch, sample_b, threads, timed_steps=3):
    """BASELINE.md §3 protocol: the oracle's full training step (fwd + RNN-T loss + bwd + AdamW, fp32, the MATERIALISING
    joint of networks/transducer.py:58-69) on `sample_b` utterances of the same workload on the host cores:
    1 warm-up step + `timed_steps` timed steps, median reported."""
    from oracle.rnnt_oracle import training_loss
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    oracle = _oracle_for(model, tn, pn, V, double=False)
    opt = torch.optim.AdamW(oracle.parameters(), lr=1e-3, weight_decay=1e-4)
    sub = tuple((x[:sample_b].cpu() if isinstance(x, torch.Tensor) else x[:sample_b]) for x in batch)
    times = []
    for _ in range(1 + timed_steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = training_loss(oracle, sub)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        progress(f"cpu_baseline step {len(times)}/{1 + timed_steps}: {times[-1]:.1f} s")
    timed = sorted(times[1:])
    med = timed[len(timed) // 2]
    return sample_b / med, med, times


def parity_vs_float64_oracle(model, tn, pn, V, batch, nb, threads):
    """Loss + the three probe gradients (SURVEY §8d "Loss delta") of the HIP path on the CURRENT weights, dropout off,
    against the float64 oracle on the first `nb` utterances of the bench batch."""
    from oracle.rnnt_oracle import training_loss
    torch.set_num_threads(threads)
    was_training = model.training
    model.eval()
    sub = tuple((x[:nb] if isinstance(x, torch.Tensor) else x[:nb]) for x in batch)
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()
    hip_l = model.jointnet.loss(sub[0], sub[2], sub[3], sub[5], sub[6], model.blank_token_id).mean()
    hip_l.backward()
    hip_loss = float(hip_l.detach())
    hip_grads = {k: p.grad.detach().double().cpu().clone() for k, p in model.jointnet.named_parameters() if k in GRAD_PROBES}
    model.train(was_training)
    oracle = _oracle_for(model, tn, pn, V, double=True)
    cpu_sub = tuple((x.cpu() if isinstance(x, torch.Tensor) else x) for x in sub)
    ref = training_loss(oracle, (cpu_sub[0].double(),) + cpu_sub[1:])
    ref.backward()
    ref_loss = float(ref.detach())
    progress(f"parity: float64 oracle done (loss {ref_loss:.6f}, HIP {hip_loss:.6f})")
    ref_grads = {k: p.grad for k, p in oracle.named_parameters() if k in GRAD_PROBES}
    # the same sample through the fp32 oracle (torch-CPU fp32 = the REFERENCE's own arithmetic): how far plain fp32 sits
    # from float64 on these gradients is the yardstick for the HIP path's distance (sums with heavy cancellation:
    # tools/parity_probe.py shows the distance does not move with exact cell math or f32-input MFMA)
    o32 = _oracle_for(model, tn, pn, V, double=False)
    l32 = training_loss(o32, cpu_sub)
    l32.backward()
    progress("parity: fp32 oracle done")
    f32_grads = {k: p.grad.double() for k, p in o32.named_parameters() if k in GRAD_PROBES}
    devs = {}
    for k in GRAD_PROBES:
        scale = float(ref_grads[k].abs().max())
        devs[k] = {"max_abs_dev": float((hip_grads[k] - ref_grads[k]).abs().max()), "ref_max_abs": scale,
                   "fp32_oracle_max_abs_dev": float((f32_grads[k] - ref_grads[k]).abs().max())}
        devs[k]["rel_to_max"] = devs[k]["max_abs_dev"] / max(scale, 1e-30)
        devs[k]["fp32_oracle_rel_to_max"] = devs[k]["fp32_oracle_max_abs_dev"] / max(scale, 1e-30)
    return abs(hip_loss - ref_loss) / abs(ref_loss), devs, abs(float(l32.detach()) - ref_loss) / abs(ref_loss)


def grad_within_tolerance(v):
    """max_abs_dev <= 2e-4 * max(|ref|max, 1e-3) (the bound tests/test_gpu_model.py uses at initial weights), or — on
    weights where plain fp32 itself is further than that from float64 — within 1.5x of the fp32 oracle's own deviation."""
    return v["max_abs_dev"] <= max(2e-4 * max(v["ref_max_abs"], 1e-3), 1.5 * v["fp32_oracle_max_abs_dev"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=list(CONFIGS))
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--ragged", action="store_true", help="KsponSpeech-shaped ragged lengths (SURVEY §8d c4 variant)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=4, help="utterances in the CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = every core this process may use (measured)")
    ap.add_argument("--parity-sample", type=int, default=2, help="utterances in the float64 parity leg")
    ap.add_argument("--profile-every", type=int, default=5,
                    help="the library's HIP-event profiler (2 events around every kernel: +0.56 ms on a 25 ms c2 step when it is on "
                         "for all of them) records every n-th step of the timed region; 1 = every step")
    ap.add_argument("--unprofiled-steps", type=int, default=0,
                    help="diagnostic: after the timed region, time this many more steps with the library's HIP-event profiler off "
                         "(reported as ms_per_step_unprofiled; never the headline value)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists in rnntransducer_amd)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from rnntransducer_amd import _lib
    from rnntransducer_amd.data import synthetic_batch

    cfg = CONFIGS[a.config]
    B, T, U, V = cfg[:4]
    model, tn, pn = build_model(cfg, a.dropout, max(100, a.warmup + a.steps + a.unprofiled_steps + 1))
    model = model.to(dev).train()
    t_lengths = None
    if a.ragged:
        # ONE global batch of world*B ragged utterances, sorted by length (descending) and dealt
        # indices[rank::world] (the behaviour of the reference's datasampler.py:74-99), so every rank sees a similar T profile
        from rnntransducer_amd.data import global_ragged_lengths, length_grouped_indices
        glob = global_ragged_lengths(world * B, T)
        t_lengths = [glob[i] for i in length_grouped_indices(glob, rank, world)]
    batch = synthetic_batch(B, T, U, V, ragged=a.ragged, seed=1234 + rank, device=dev, t_lengths=t_lengths)
    conf = model.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    # configure_optimizers() on a GPU module returns FlatAdamW: .grad tensors are views of ONE flat buffer, so the DP
    # exchange is one RCCL all-reduce and the update one fused kernel

    def step():
        opt.zero_grad()
        loss = model.training_step(batch, 0)["loss"]
        loss.backward()
        opt.all_reduce_grads()
        opt.step()
        sched.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    L = _lib.lib()
    every = max(1, a.profile_every)
    nprof = len(range(0, a.steps, every))   # steps of the timed region whose kernels are timed by HIP events
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        L.rnnt_hip_prof_enable(1 if i % every == 0 else 0)
        loss = step()
    L.rnnt_hip_prof_enable(0)
    fence()
    dt = time.perf_counter() - t0
    last_loss = float(loss.detach())
    dt_unprof = None
    if a.unprofiled_steps > 0:
        fence()
        t0 = time.perf_counter()
        for _ in range(a.unprofiled_steps):
            step()
        fence()
        dt_unprof = (time.perf_counter() - t0) / a.unprofiled_steps
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-kernel totals over the timed region (HIP events on the launch stream, recorded inside the library)
    nk = len(_lib.KERNEL_KINDS)
    ms = (ctypes.c_double * nk)()
    work = (ctypes.c_double * nk)()
    cnt = (ctypes.c_int64 * nk)()
    _lib.check(L.rnnt_hip_prof_collect(ms, work, cnt, nk), "prof_collect")
    kernels = {}
    for i, name in enumerate(_lib.KERNEL_KINDS):
        if cnt[i]:
            kernels[name] = {"launches": int(cnt[i]), "ms_total": round(ms[i], 3), "avg_us": round(1e3 * ms[i] / cnt[i], 2),
                             "work_per_launch": work[i] / cnt[i]}
    # GEMM arithmetic (include/rnnt_hip.h).  The big products of the LSTM layers run on gemm_hp_kernel: fp32 operands as two
    # fp16 pieces ("half pair", per-row power-of-two scale), 3 piece products on v_mfma_f32_16x16x32_f16, fp32 accumulate:
    # roofline = f16 dense peak / 3.  Everything else runs on gemm.hip: default 3 bf16 pieces / 6 products (peak / 6),
    # "f32" = the f32-input MFMA.
    gemm_mode = os.environ.get("RNNT_GEMM_MODE") or "bf16x6"
    nprod = {"bf16x6": 6, "bf16x3": 3}.get(gemm_mode, 1)
    gemm_name = "gemm_f32_kernel" if nprod == 1 else "gemm_bf16s_kernel"  # incl. its 256x256-tile form gemm_bf16s256_kernel
    gemm_peak = {gemm_name: PEAK_FP32_MFMA_TFLOPS if nprod == 1 else PEAK_BF16_MFMA_TFLOPS / nprod,
                 "gemm_hp_kernel": PEAK_BF16_MFMA_TFLOPS / 3}
    if nprod > 1 and "gemm_f32_kernel" in kernels:  # the profiler's kind 0 is "the gemm.hip kernel", whichever form ran
        kernels = {(gemm_name if k == "gemm_f32_kernel" else k): v for k, v in kernels.items()}
    for name, kd_ in kernels.items():  # every kernel kind against its own roofline (GEMMs: MFMA; the rest: HBM)
        sec = kd_["ms_total"] / kd_["launches"] / 1e3
        if name in gemm_peak:
            kd_["tflops"] = round(kd_["work_per_launch"] / sec / 1e12, 2)  # algorithmic 2MNK, fp32-equivalent
            kd_["frac_of_mfma_peak"] = round(kd_["tflops"] / gemm_peak[name], 4)
        else:
            kd_["gbs"] = round(kd_["work_per_launch"] / sec / 1e9, 1)
            kd_["frac_of_hbm_peak"] = round(kd_["gbs"] / PEAK_HBM_GBS, 5)
        kd_["ms_per_step"] = round(kd_["ms_total"] / nprof, 3)
    dom = max((k for k in kernels if k not in ("misc", "hp_split_kernels")), key=lambda k: kernels[k]["ms_total"])
    kd = kernels[dom]
    per_launch_s = kd["ms_total"] / kd["launches"] / 1e3
    if dom in gemm_peak:
        achieved = kd["work_per_launch"] / per_launch_s / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": round(gemm_peak[dom], 1),
                "unit": "TFLOP/s", "frac": round(achieved / gemm_peak[dom], 4), "traffic": None}
        roof["note"] = ("fp32-equivalent FLOP/s; peak = 2500 dense f16/bf16 MFMA TFLOP/s divided by the MFMA products one fp32 product "
                        "costs (3 on half-pair f16 operands, 6 on bf16 pieces); the f32-input MFMA peak is 157.3")
    else:
        achieved = kd["work_per_launch"] / per_launch_s / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(achieved / PEAK_HBM_GBS, 5), "traffic": None}
        if dom.startswith("lstm_"):
            # the kind's launches: one per encoder layer (T dependent timesteps) and one per prediction-net layer (U + 1)
            timesteps = nprof * (cfg[4][1] * T + cfg[5][1] * (U + 1))
            roof["note"] = (f"persistent recurrence: {cfg[4][1]} launches of {T} dependent timesteps + {cfg[5][1]} of {U + 1} per step, "
                            f"{1e3 * kd['ms_total'] / timesteps:.2f} us per "
                            "timestep; bounded by the per-step exchange (one L2 hand-off) + 48 MFMAs + cell math chain, not by HBM: "
                            "algorithmic bytes/launch = gates r+w, c, y or dy per (t,b,d) + weights (DESIGN.md section 4)")
    # HBM bytes per launch from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as
    # MI355X_MICROARCH.md prescribes); collected offline because counters cannot be read from inside the process
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
        if pmc.get("config") == a.config:
            roof["traffic"] = round(pmc["hbm_bytes_per_launch"][dom])
            roof["traffic_source"] = pmc["from"]
    except (OSError, KeyError, ValueError):
        pass
    roof["avg_launch_us"] = kd["avg_us"]
    roof["launches"] = kd["launches"]
    roof["profiled_steps"] = f"{nprof} of the {a.steps} timed steps (every {every}th; HIP events on the launch stream around every kernel)"

    out = {
        "metric": "utterances/sec", "value": round(world * B * a.steps / dt, 3), "unit": "utt/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "gemm_arithmetic": ("f16x3 half-pair operands (LSTM input projection, dX, dW_ih, dW_hh) + " if not os.environ.get("RNNT_GEMM_NO_HP") else "") + gemm_mode,
        "recurrence_arithmetic": "f32 MFMA (v2 kernels)" if os.environ.get("RNNT_LSTM_V2") else
                                 ("bf16x6 (v3/v4)" if os.environ.get("RNNT_LSTM_NO_V5") else "f16x3 half-pair operands, fp32 accumulate (v5; H > 512: bf16x6 v3/v4)"),
        "data": "synthetic",
        "config": {"workload": f"{'BASELINE configs[%d]' % (list(CONFIGS).index(a.config) + (1 if a.config == 'c5' else 0)) if a.config != 'shipped' else 'reference config.json'} {a.config}: full train step, B={B}/GPU T={T} "
                               f"(10 ms frames x 80 mel) U={U} V={V}, enc {cfg[4][1]}x{cfg[4][0]} bi-{tn['rnn_type'].upper()}, pred {cfg[5][1]}x{cfg[5][0]} {pn['rnn_type'].upper()}, "
                               f"O={cfg[6]}, dropout {a.dropout}, {'ragged' if a.ragged else 'fixed'} lengths",
                   "global_batch": world * B, "parallelism": f"dp{world}", "grad_allreduce_bytes": opt.grad_bytes()},
        "last_loss": round(last_loss, 4), "roofline": roof, "kernels": kernels,
    }
    # the step must run without any (B,T,U+1,V) or (B,T,U+1,2*O) tensor; peak allocator bytes over the
    # timed region vs what the reference's joint materialises (networks/transducer.py:61-69)
    O = cfg[6]
    out["memory"] = {"peak_allocated_bytes": int(torch.cuda.max_memory_allocated()),
                     "reference_concat_bytes_per_copy": int(B) * T * (U + 1) * 2 * O * 4,
                     "reference_logits_bytes": int(B) * T * (U + 1) * V * 4,
                     "stash_note": "peak is dominated by the LSTM stash (activated gates, 16*H bytes per frame per direction per layer)"}

    if dt_unprof is not None:
        out["ms_per_step_unprofiled"] = round(1e3 * dt_unprof, 3)
    parity_ok = True
    if rank == 0:
        progress(f"timed region done: {out['value']} utt/s, {out['ms_per_step']} ms per step")
    if world == 1 and not a.no_cpu_baseline:
        cores, cpu_model = host_cpu_info()
        threads = a.cpu_threads or cores
        # (1) parity: HIP (dropout off, CURRENT weights — they moved during the timed steps) vs the FLOAT64 oracle
        npar = min(a.parity_sample, B)
        out["loss_rel_delta"], out["grad_max_abs_dev"], f32_loss_rel = parity_vs_float64_oracle(model, tn, pn, V, batch, npar, threads)
        out["parity"] = {"oracle": f"float64 CPU oracle, first {npar} utterances of the bench batch, dropout off, weights after "
                                   f"{a.warmup + a.steps} AdamW steps",
                         "loss_rel_tol": 1e-4, "fp32_oracle_loss_rel_delta": f32_loss_rel,
                         "grad_tol": "max_abs_dev <= max(2e-4 * max(ref_max_abs, 1e-3), 1.5 * fp32_oracle_max_abs_dev): within the "
                                     "fixture tests' bound, or as close to float64 as the reference's own fp32 arithmetic (torch CPU)"}
        parity_ok = out["loss_rel_delta"] <= 1e-4 and all(grad_within_tolerance(v) for v in out["grad_max_abs_dev"].values())
        out["parity"]["ok"] = parity_ok
        # (2) reported CPU baseline: fp32 oracle, 1 warm-up + 3 timed full train steps, median
        nb = min(a.cpu_sample, B)
        v, med, times = cpu_baseline(model, tn, pn, V, batch, nb, threads)
        out["cpu_baseline"] = {"value": round(v, 4), "unit": "utt/s", "cores": threads, "kind": "port",
                               "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
                               "sample": f"full train steps (fwd + RNN-T loss + bwd + AdamW, fp32, materialising joint) of the oracle on "
                                         f"{nb} utterances of the same workload: 1 warm-up ({times[0]:.1f} s) + {len(times) - 1} timed, median "
                                         f"{med:.1f} s/step, torch.set_num_threads({threads})"}
        out["speedup_vs_cpu_baseline"] = round(out["value"] / v, 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if not parity_ok:
        raise SystemExit("bench.py: parity leg outside tolerance (see loss_rel_delta / grad_max_abs_dev in the line above)")


if __name__ == "__main__":
    main()

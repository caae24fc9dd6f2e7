#!/usr/bin/env python3
"""bench.py — utterances/sec of the full RNN-T training step on the MI355X HIP hot path.

    python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 without WORLD_SIZE in the environment the
                                                              process is a LAUNCHER — before any GPU call it starts N ranks
                                                              through torch.distributed.run on 127.0.0.1, forwards rank 0's
                                                              JSON line and exits with the ranks' code; fewer than N visible
                                                              devices is an error, never a silent 1-GPU run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                (N>1, one rank per GPU, RCCL over xGMI: what the launcher runs)

A "step" = forward (LSTM encoder + LSTM prediction net) + fused joint/RNN-T loss + backward + flat-gradient
all-reduce + AdamW, on one synthetic batch that is already resident in HBM.  Workload at every N: BASELINE.json
configs[1] per GPU — B=32, T=1000 (10 s @ 80 mel), U=40, V=72, 4x512 bi-LSTM encoder / 1x512 LSTM prediction net,
fp32, inter-layer dropout 0.2.  Weak scaling: per-GPU work fixed; value = total utterances / s.

Rank 0 prints ONE JSON line with the driver's contract plus
  "roofline"     for the kernel that took the most time inside the timed region (live HIP-event timing from the
                 library's opt-in profiler; algorithmic FLOPs/bytes per launch are computed in the launch wrappers),
  "cpu_baseline" the CPU oracle (torch-CPU composite of the reference path + C loss) timed on this box's host cores
                 on a bounded sample, N=1 only,
  "loss_rel_delta" / "grad_max_abs_dev"  the HIP path against the FLOAT64 oracle at the INITIAL weights, measured BEFORE the
                 timed steps on the launch geometry that is timed: the HIP side runs the FULL bench batch (B rows, the same sync
                 groups / workgroup count as a timed step) with per-utterance losses, and back-propagates the mean over the first
                 `--parity-sample` utterances only (the other rows get upstream weight 0; rows are independent), so loss and
                 gradients are comparable with the oracle's run on those utterances.  Gate: loss 1e-4 relative, gradients 2e-4 of
                 the tensor's maximum — the fixture tests' bound, no widening.  The same comparison on the weights the timed steps
                 left is reported under "parity_after_training" (informational: conditioning, not a gate).
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


def parity_vs_float64_oracle(model, opt, tn, pn, V, batch, nb, threads, with_fp32_oracle):
    """Loss + the three probe gradients (SURVEY §8d "Loss delta") of the HIP path on the CURRENT weights, dropout off, against the
    float64 oracle on the first `nb` utterances of the bench batch.  The HIP side runs the WHOLE bench batch — the launch that is
    timed: same B, same sync groups, same workgroup count — with reduction "none" and back-propagates nll[:nb].mean(): rows >= nb
    get upstream weight 0 and, rows being independent, contribute nothing to any gradient."""
    from oracle.rnnt_oracle import training_loss
    torch.set_num_threads(threads)
    was_training = model.training
    model.eval()
    opt.zero_grad()
    nll = model.jointnet.loss(batch[0], batch[2], batch[3], batch[5], batch[6], model.blank_token_id, reduction="none",
                              audio_lengths=batch[1])   # with the host list, as training_step: ragged batches take the valid-frame plan
    hip_l = nll[:nb].mean()
    hip_l.backward()
    hip_loss = float(hip_l.detach())
    hip_nll = nll.detach().double().cpu()
    hip_grads = {k: p.grad.detach().double().cpu().clone() for k, p in model.jointnet.named_parameters() if k in GRAD_PROBES}
    opt.zero_grad()
    model.train(was_training)
    oracle = _oracle_for(model, tn, pn, V, double=True)
    cpu_sub = tuple((x[:nb].cpu() if isinstance(x, torch.Tensor) else x[:nb]) for x in batch)
    # the checker runs its LSTMs one utterance at a time on its own valid prefix instead of through a PackedSequence: the same function
    # (tests/test_oracle_networks.py::test_per_utterance_lstm_equals_the_packed_one, 1e-12) without the O(T^2) CPU autograd of packed
    # batches (config 3: 114 s -> 15 s).  The timed cpu_baseline below keeps the packed path: that is what the reference runs.
    from oracle import rnnt_oracle as _ro
    _ro.PER_UTTERANCE = True
    try:
        ref = training_loss(oracle, (cpu_sub[0].double(),) + cpu_sub[1:])
        ref.backward()
    finally:
        _ro.PER_UTTERANCE = False
    ref_loss = float(ref.detach())
    progress(f"parity: float64 oracle done (loss {ref_loss:.6f}, HIP {hip_loss:.6f}; HIP ran all {len(hip_nll)} rows)")
    ref_grads = {k: p.grad for k, p in oracle.named_parameters() if k in GRAD_PROBES}
    f32_grads, f32_loss_rel = None, None
    if with_fp32_oracle:
        # the same sample through the fp32 oracle (torch-CPU fp32 = the REFERENCE's own arithmetic): how far plain fp32 sits
        # from float64 on these gradients — informational yardstick for the post-training comparison
        o32 = _oracle_for(model, tn, pn, V, double=False)
        l32 = training_loss(o32, cpu_sub)
        l32.backward()
        progress("parity: fp32 oracle done")
        f32_grads = {k: p.grad.double() for k, p in o32.named_parameters() if k in GRAD_PROBES}
        f32_loss_rel = abs(float(l32.detach()) - ref_loss) / abs(ref_loss)
    devs = {}
    for k in GRAD_PROBES:
        scale = float(ref_grads[k].abs().max())
        devs[k] = {"max_abs_dev": float((hip_grads[k] - ref_grads[k]).abs().max()), "ref_max_abs": scale}
        devs[k]["rel_to_max"] = devs[k]["max_abs_dev"] / max(scale, 1e-30)
        if f32_grads is not None:
            devs[k]["fp32_oracle_max_abs_dev"] = float((f32_grads[k] - ref_grads[k]).abs().max())
            devs[k]["fp32_oracle_rel_to_max"] = devs[k]["fp32_oracle_max_abs_dev"] / max(scale, 1e-30)
    return abs(hip_loss - ref_loss) / abs(ref_loss), devs, f32_loss_rel, bool(torch.isfinite(hip_nll).all())


def grad_within_tolerance(v):
    """max_abs_dev <= 2e-4 * max(|ref|max, 1e-3): the bound tests/test_gpu_model.py and tests/test_gpu_configs.py use."""
    return v["max_abs_dev"] <= 2e-4 * max(v["ref_max_abs"], 1e-3)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a, argv):
    """`python bench.py --gpus N` (N > 1) without a distributed environment: this process becomes the launcher.  It touches no GPU
    (torch.cuda.device_count() does not initialise one), checks that N devices are visible, starts N ranks — one per GPU — through
    `python -m torch.distributed.run` on 127.0.0.1 (what scripts/run_train.sh:7-9 of the reference does with torchrun), lets their
    stdout / stderr through (rank 0 prints the JSON line) and exits with the ranks' return code."""
    import subprocess
    if a.share_gpu and a.backend != "gloo":
        raise SystemExit("--share-gpu needs --backend gloo (RCCL does not take two ranks on one device)")
    if not a.stub and not a.share_gpu:
        ndev = torch.cuda.device_count()
        if ndev < a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) are visible; refusing to measure fewer devices than asked")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")              # scripts/run_train.sh:7
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    progress(f"launcher: starting {a.gpus} ranks: {' '.join(cmd[1:8])} ...")
    rc = subprocess.call(cmd, env=env)
    raise SystemExit(rc)


def run_stub(a, rank, world):
    """Launcher / sharding / timing plumbing on the CPU (tests only: `--stub`, gloo, no model, no kernel): everything the N > 1 path
    of main() does AROUND the step — process group, the c4 ragged deal, the barrier-bracketed timed loop with the flat-gradient
    all-reduce (the product's FlatParams on CPU tensors), the max-over-ranks reduction — with the step itself replaced by a sleep.
    Prints a line that cannot be mistaken for a measurement (metric "stub")."""
    from rnntransducer_amd.data import global_ragged_lengths, length_grouped_indices, synthetic_batch
    from rnntransducer_amd.optim import FlatParams
    if world > 1:
        dist.init_process_group("gloo")
    B, T, U, V = CONFIGS[a.config][:4]
    t_lengths = None
    if a.ragged:
        glob = global_ragged_lengths(world * B, T)
        t_lengths = [glob[i] for i in length_grouped_indices(glob, rank, world)]
    batch = synthetic_batch(B, T, U, V, ragged=a.ragged, seed=1234 + rank, device="cpu", t_lengths=t_lengths)
    torch.manual_seed(0)
    flat = FlatParams(torch.nn.Linear(8, 8).parameters())
    status = torch.zeros(4, dtype=torch.int32)

    def step():
        flat.zero_grad()
        flat.flat_grad[:flat.n_param] += float(rank + 1)
        time.sleep(0.002 * (rank + 1))
        flat.all_reduce_grads(status)

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    own_dt = dt
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    out = {"metric": "stub (launcher test only, nothing was measured)", "value": None, "unit": "utt/s", "n_gpus": world,
           "rccl_ranks": dist.get_world_size() if world > 1 else 1, "steps": a.steps, "warmup": a.warmup, "data": "stub",
           "ms_per_step": round(1e3 * dt / a.steps, 3), "own_ms_per_step": round(1e3 * own_dt / a.steps, 3),
           "grad_sum_per_element": float(flat.flat_grad[0]), "status_slot": float(flat.status_slot()[0]),
           "config": {"workload": f"stub {a.config}", "global_batch": world * B, "parallelism": f"dp{world}",
                      "rank_t_lengths_head": batch[1][:4], "ragged": bool(a.ragged)}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


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
    ap.add_argument("--stub", action="store_true",
                    help="tests only: run the launcher / sharding / timing plumbing on the CPU (gloo) with the step replaced by a "
                         "sleep; prints metric 'stub', never a measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N > 1 path (nccl = RCCL over xGMI: the measured configuration; gloo: rehearsals)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --backend gloo), so the N > 1 code path — model step, flat all-reduce "
                         "with the status slot, barrier, max-over-ranks timing — runs end to end on a one-GPU box; the line says so and is "
                         "not a measurement of N GPUs")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a, sys.argv[1:])   # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the two must agree (start bench.py without a distributed "
                         f"environment and it launches its own {a.gpus} ranks)")
    if a.stub:
        return run_stub(a, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists in rnntransducer_amd)")
    if a.share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

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
        if a.share_gpu and world > 1:
            # rehearsal on one device: the ranks take turns for forward + backward (two processes' persistent recurrences cannot be
            # co-resident on one GPU); everything after it — the exchange, the guarded update — is the N > 1 path as it ships
            for turn in range(world):
                if turn == rank:
                    loss = model.training_step(batch, 0)["loss"]
                    loss.backward()
                    torch.cuda.synchronize()
                dist.barrier()
        else:
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

    # parity gate FIRST: initial weights, the full bench batch on the HIP side (the launch geometry that is timed), float64 oracle
    # on the first --parity-sample utterances, strict tolerances (loss 1e-4 relative, gradients 2e-4 of the tensor's maximum)
    parity = None
    want_cpu_legs = world == 1 and not a.no_cpu_baseline
    if want_cpu_legs:
        cores, cpu_model = host_cpu_info()
        threads = a.cpu_threads or cores
        npar = min(a.parity_sample, B)
        lrd, gdev, _, finite = parity_vs_float64_oracle(model, opt, tn, pn, V, batch, npar, threads, with_fp32_oracle=False)
        parity = {"loss_rel_delta": lrd, "grad_max_abs_dev": gdev, "all_rows_finite": finite,
                  "ok": bool(finite and lrd <= 1e-4 and all(grad_within_tolerance(v) for v in gdev.values()))}

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
        from rnntransducer_amd.csrc.build import source_digest
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
        # only numbers collected on THIS build of the kernels (the file is stamped with a digest of csrc/): stale ones are omitted
        if pmc.get("config") == a.config and pmc.get("csrc_sha16") == source_digest():
            roof["traffic"] = round(pmc["hbm_bytes_per_launch"][dom])
            roof["traffic_source"] = pmc["from"]
        elif pmc.get("config") == a.config:
            roof["traffic_note"] = "profiles/pmc_traffic_latest.json was collected on another build of csrc/: omitted"
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
    out["rccl_ranks"] = dist.get_world_size() if world > 1 else 1
    if world > 1 and (a.backend != "nccl" or a.share_gpu):
        out["rehearsal"] = f"backend {a.backend}, {'all ranks on cuda:0' if a.share_gpu else 'one GPU per rank'}: NOT a measurement of {world} GPUs over RCCL"
        out["metric"] = "utterances/sec (rehearsal of the N > 1 path, not a multi-GPU measurement)"
    parity_ok = True
    if rank == 0:
        progress(f"timed region done: {out['value']} utt/s, {out['ms_per_step']} ms per step")
    if want_cpu_legs:
        # (1) the parity gate measured before the timed steps (initial weights, full-batch HIP launch, float64 oracle)
        out["loss_rel_delta"], out["grad_max_abs_dev"] = parity["loss_rel_delta"], parity["grad_max_abs_dev"]
        parity_ok = parity["ok"]
        out["parity"] = {"oracle": f"float64 CPU oracle on the first {npar} utterances of the bench batch, INITIAL weights, dropout off; the "
                                   f"HIP side ran all {B} rows of the bench batch (the timed launch geometry) with upstream weight 0 on "
                                   f"rows >= {npar}", "loss_rel_tol": 1e-4,
                         "grad_tol": "max_abs_dev <= 2e-4 * max(ref_max_abs, 1e-3) (the fixture tests' bound)",
                         "all_rows_finite": parity["all_rows_finite"], "ok": parity_ok}
        # (2) informational: the same comparison on the weights the timed steps left, next to the fp32 oracle's own distance from
        # float64 (sums with heavy cancellation: conditioning, tools/parity_probe.py) — reported, never a gate
        lrd2, gdev2, f32_loss_rel, finite2 = parity_vs_float64_oracle(model, opt, tn, pn, V, batch, npar, threads, with_fp32_oracle=True)
        out["parity_after_training"] = {"weights": f"after {a.warmup + a.steps} AdamW steps", "informational": True,
                                        "loss_rel_delta": lrd2, "fp32_oracle_loss_rel_delta": f32_loss_rel,
                                        "grad_max_abs_dev": gdev2, "all_rows_finite": finite2}
        # (3) reported CPU baseline: fp32 oracle, 1 warm-up + 3 timed full train steps, median
        nb = min(a.cpu_sample, B)
        v, med, times = cpu_baseline(model, tn, pn, V, batch, nb, threads)
        out["cpu_baseline"] = {"value": round(v, 4), "unit": "utt/s", "cores": threads, "kind": "port",
                               "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(),
                               "sample": f"full train steps (fwd + RNN-T loss + bwd + AdamW, fp32, materialising joint) of the oracle on "
                                         f"{nb} utterances of the same workload: 1 warm-up ({times[0]:.1f} s) + {len(times) - 1} timed, median "
                                         f"{med:.1f} s/step, torch.set_num_threads({threads})"}
        out["speedup_vs_cpu_baseline"] = round(out["value"] / v, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not parity_ok:
        raise SystemExit("bench.py: parity gate outside tolerance (see loss_rel_delta / grad_max_abs_dev in the line above)")


if __name__ == "__main__":
    main()

"""Data-parallel path with REAL kernels at world 2 on the one GPU the test box has: two processes share cuda:0 and
take turns for their forward + backward (two processes' persistent recurrences cannot be co-resident on one device), exchange the flat
gradient buffer through the product's own `FlatAdamW.all_reduce_grads()` (gloo on device tensors here; RCCL on a node with one GPU per
rank) and take the fused AdamW step.  Checked against ONE process training on the global batch: DDP semantics (train.py:45) — the
average of the per-rank mean-loss gradients equals the gradient of the global mean loss (equal shard sizes) — and identical
parameters on both ranks after the update."""
import os
import socket
from argparse import Namespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
V = 30
ARGS = dict(learning_rate=1e-3, weight_decay=1e-4, warmup_ratio=0.2, final_div_factor=1e4, total_steps=100, move_metrics_to_cpu=False)


def _model():
    from rnntransducer_amd import RNNTransducer
    torch.manual_seed(11)
    tn = dict(input_size=80, hidden_size=128, output_size=64, num_layers=2, dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=V, hidden_size=64, output_size=64, num_layers=1, dropout=0.0)
    return RNNTransducer(pn, tn, dict(num_classes=V), Namespace(**ARGS)).cuda().train()


def _global_batch():
    from rnntransducer_amd.data import synthetic_batch
    return synthetic_batch(8, 60, 9, V, ragged=True, seed=17, device="cpu")


def _shard(batch, rank, world):
    n = batch[0].shape[0] // world
    sl = slice(rank * n, (rank + 1) * n)
    return tuple((x[sl].cuda() if isinstance(x, torch.Tensor) else x[sl]) for x in batch)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model = _model()
    opt = model.configure_optimizers()["optimizer"]          # FlatAdamW with direct flat gradients (no trainer attached)
    assert opt.world == world
    batch = _shard(_global_batch(), rank, world)
    opt.zero_grad()
    for turn in range(world):                                 # one rank's recurrences on the device at a time
        if turn == rank:
            model.training_step(batch, 0)["loss"].backward()
            torch.cuda.synchronize()
        dist.barrier()
    opt.all_reduce_grads()                                    # ONE collective over the flat buffer (+ the status slot), SUM
    grads = {k: (p.grad * opt._grad_scale).detach().cpu() for k, p in model.named_parameters()}
    slot = float(opt.flat.status_slot()[0])
    opt.step()                                                # fused update: 1/world folded in, guarded by the collective's status slot
    torch.cuda.synchronize()
    out[rank] = (grads, {k: p.detach().cpu() for k, p in model.named_parameters()}, slot)
    dist.destroy_process_group()


def test_world2_flat_allreduce_and_update_equal_one_process_on_the_global_batch():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_rank, args=(world, port, out), nprocs=world, join=True)
        (g0, p0, s0), (g1, p1, s1) = out[0], out[1]
    assert s0 == 0.0 and s1 == 0.0                            # no rank's recurrences gave up
    for k in g0:
        assert torch.equal(g0[k], g1[k]) and torch.equal(p0[k], p1[k]), k      # both ranks hold the same sums and the same parameters
    # one process, the global batch of 8 (mean over 8 = average of the two shards' means)
    model = _model()
    opt = model.configure_optimizers()["optimizer"]
    batch = _shard(_global_batch(), 0, 1)
    opt.zero_grad()
    model.training_step(batch, 0)["loss"].backward()
    ref_g = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        scale = max(ref_g[k].abs().max().item(), 1e-3)
        assert (g0[k] - ref_g[k]).abs().max().item() < 1e-5 * scale, k
        assert (p0[k] - p.detach().cpu()).abs().max().item() < 2e-6, k

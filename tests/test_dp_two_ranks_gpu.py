"""GPU, world_size 2: the data-parallel training step with the REAL engine (HIP kernels), the real Trainer (bucket launches
from inside backward, per-layer Adam slices on the side stream that wait for their bucket) and two ranks.

Only one GPU is available to these tests and RCCL refuses two ranks on one device, so the two rank processes share cuda:0
and the process group is gloo -- torch's gloo backend all-reduces device tensors (staged through the host), with the same
asynchronous Work / wait() contract the Trainer uses over RCCL.  What this pins that no other test does: >1 rank + device
kernels + the overlap machinery together (tests/test_dp_gloo.py has >1 rank but an oracle engine on CPU;
tests/test_dp_rccl_gpu.py has RCCL but one rank).

Expected values: a single-process emulation of SURVEY.md §8(e) -- two replicas with per-replica BatchNorm statistics, each
backward on its own shard, gradients summed, Adam with grad_scale = 1/2 -- built from the same kernels.  A two-term sum is
order-independent, so the two-rank run must match it BIT FOR BIT, and both ranks must hold identical parameters."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phasegen import detgen

pytestmark = pytest.mark.gpu

C, L, BPER, WORLD, STEPS = 16, 64, 2, 2, 3


def shard(rank):
    return torch.from_numpy(detgen.make_batch(BPER, C, L, seed=1 + rank))          # bench.py's per-rank seeding


def worker(rank, port, q, compress):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from phasegen.model import UNetModel
        from phasegen.trainer import Trainer
        torch.cuda.set_device(0)
        model = UNetModel(C, 2 * C).load_numpy(detgen.make_params(C, seed=0))
        tr = Trainer(model, grad_compress=compress)
        assert tr.world == WORLD and tr.overlap_adam and model.engine.contended
        batch = shard(rank).cuda()
        losses = []
        for _ in range(STEPS):
            losses.append(tr.step(batch).cpu().numpy().copy())
            assert not tr.reducer.pending
        torch.cuda.synchronize()
        q.put((rank, np.stack(losses), model.engine.arena.flat.cpu().numpy().copy(),
               model.engine.arena.grad.cpu().numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def emulate(compress):
    """Two replicas in one process: per-replica forward / loss / backward, gradient sum, Adam with grad_scale 1/2."""
    from phasegen.model import UNetModel
    from phasegen.trainer import Trainer
    trs = []
    for _ in range(WORLD):
        m = UNetModel(C, 2 * C).load_numpy(detgen.make_params(C, seed=0))
        m.engine.contended = True                                   # the work split a data-parallel rank uses
        trs.append(Trainer(m, overlap_adam=False))
    batches = [shard(r).cuda() for r in range(WORLD)]
    losses = [[] for _ in range(WORLD)]
    for _ in range(STEPS):
        for tr, b in zip(trs, batches):
            pred = tr.engine.forward(b[:, 0])
            dpred = torch.empty_like(pred)
            tr._loss(pred, b, dpred, tr.losses, tr.mag_weight)
            tr.engine.backward(dpred, lambda name: None)
        grads = [tr.engine.arena.grad for tr in trs]
        if compress == "bf16":                                      # the wire format: each rank's bucket rounded to bf16, summed in bf16
            total = (grads[0].to(torch.bfloat16) + grads[1].to(torch.bfloat16)).float()
        else:
            total = grads[0] + grads[1]
        for r, tr in enumerate(trs):
            losses[r].append(tr.losses.cpu().numpy().copy())
            tr.engine.arena.grad.copy_(total)
            tr.optim.step(grad_scale=1.0 / WORLD)
    torch.cuda.synchronize()
    return [np.stack(l) for l in losses], trs[0].engine.arena.flat.cpu().numpy(), total.cpu().numpy()


@pytest.mark.parametrize("compress", [None, "bf16"])
def test_two_ranks_real_engine_match_the_two_replica_emulation(compress):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=worker, args=(r, port, q, compress)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(WORLD):
        rank, losses, flat, grad = q.get(timeout=300)
        got[rank] = (losses, flat, grad)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want_losses, want_flat, want_grad = emulate(compress)
    assert np.array_equal(got[0][1], got[1][1]), "ranks diverged"
    assert np.array_equal(got[0][2], got[1][2])
    for r in range(WORLD):
        assert np.array_equal(got[r][0], want_losses[r]), (r, got[r][0], want_losses[r])
    assert np.array_equal(got[0][2], want_grad)
    assert np.array_equal(got[0][1], want_flat)
    assert not np.array_equal(want_losses[0], want_losses[1])       # the shards really differ


def val_worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from phasegen.model import UNetModel
        from phasegen.validate import validation_metrics
        torch.cuda.set_device(0)
        model = UNetModel(C, 2 * C).load_numpy(detgen.make_params(C, seed=0))
        vb = val_batch().cuda()
        got = validation_metrics(model, vb, hop_length=8, n_fft=2 * C, gl_iters=3, gl_seed=5, shard=True)
        bufs = {k: v.cpu().numpy().tolist() for k, v in sorted(model.engine.arena.buffers.items())}
        q.put((rank, (got, bufs)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def val_batch():
    from oracle import signal_ref
    n_fft, hop, n_clips = 2 * C, 8, 5
    n = hop * (L - 1)
    clips = [detgen.make_clip(n, seed=220 + i) for i in range(n_clips)]
    P = signal_ref.get_spec_and_angle(np.stack([signal_ref.chunk_and_stft(c, n_fft, hop) for c in clips]))
    return torch.from_numpy(np.ascontiguousarray(P, dtype=np.float32))


def test_sharded_validation_equals_single_process():
    """train.py validates at a step boundary on EVERY rank: rank r takes clips r::W (5 clips over 2 ranks:
    3 + 2), the three sums are all-reduced.  Both ranks must report the single-process result (clip c's Griffin-Lim start is
    seeded gl_seed + c wherever it runs; only the order of the final sums differs)."""
    from phasegen.model import UNetModel
    from phasegen.validate import validation_metrics
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=val_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model = UNetModel(C, 2 * C).load_numpy(detgen.make_params(C, seed=0))
    want = validation_metrics(model, val_batch().cuda(), hop_length=8, n_fft=2 * C, gl_iters=3, gl_seed=5)
    assert got[0][0] == got[1][0]
    for k in ("MSE", "NOPMSE", "LMSE"):
        assert abs(got[0][0][k] - want[k]) < 1e-6 * abs(want[k]), (k, got[0][0][k], want[k])
    # the train-mode forwards of validation update the BatchNorm buffers with each rank's own clips (3 on rank 0, 2 on rank 1):
    # rank 0's are broadcast afterwards, so the replicas stay identical -- and they are rank 0's three updates
    assert got[0][1] == got[1][1], "BatchNorm buffers diverged between the ranks during sharded validation"
    assert all(v == 3 for k, v in got[0][1].items() if k.endswith("num_batches_tracked"))

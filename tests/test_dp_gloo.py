"""CPU, world_size 2, gloo: the data-parallel path of phasegen.trainer (the same BucketedAllReduce / GradBuckets /
loader sharding code that runs over RCCL on the GPUs).

Semantics under test: nn.parallel.data_parallel computes BatchNorm statistics PER REPLICA and sums
replica gradients of a globally averaged loss == average over ranks of each rank's mean-loss gradient.  Each rank
computes its shard's gradients with the oracle (the GPU engine cannot run here), puts them in a ParamArena's gradient
arena, and runs the product's bucketed all-reduce; the result must equal the mean of the per-rank oracle gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phasegen import detgen

C, L, BPER, WORLD = 8, 24, 2, 2


def shard_grads(rank):
    from oracle import unet_ref
    p = unet_ref.to_torch(detgen.make_params(C, seed=0))
    for k in detgen.param_order():
        p[k].requires_grad_(True)
    batch = torch.from_numpy(detgen.make_batch(BPER, C, L, seed=1 + rank))      # bench.py's per-rank seeding
    out = unet_ref.unet_forward(p, batch[:, 0])
    loss, _, _ = unet_ref.phase_loss(out, batch)
    loss.backward()
    return {k: p[k].grad.detach().clone() for k in detgen.param_order()}


def worker(rank, port, q, compress=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from phasegen.trainer import BucketedAllReduce
        from phasegen.unet import BACKWARD_ORDER, ParamArena
        torch.set_num_threads(2)
        arena = ParamArena(C, torch.device("cpu"))
        g = shard_grads(rank)
        for k in detgen.param_order():
            arena.g(k).copy_(g[k])
        red = BucketedAllReduce(arena, compress=compress)
        assert red.world == WORLD and red.buckets.covers_arena()
        for name in BACKWARD_ORDER:            # the order engine.backward() calls on_grads_ready
            red.launch(name)
        assert red.wait_all() == BACKWARD_ORDER
        avg = {k: (arena.g(k) * (1.0 / WORLD)).numpy().copy() for k in detgen.param_order()}   # Adam's grad_scale
        if rank == 0:
            q.put(avg)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("compress", [None, "bf16"])
def test_bucketed_allreduce_equals_mean_of_replica_grads(compress):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, port, q, compress)) for r in range(WORLD)]
    for p in procs:
        p.start()
    avg = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = [shard_grads(r) for r in range(WORLD)]
    for k in detgen.param_order():
        w = (want[0][k] + want[1][k]).numpy() / WORLD
        # fp32 payload: exact to rounding; bf16 payload: each rank's gradient and the sum are rounded to bf16
        tol = 1e-6 if compress is None else 1.2e-2
        assert np.max(np.abs(avg[k] - w)) <= tol * max(np.max(np.abs(w)), 1e-12), k


def test_buckets_follow_backward_order_and_tile_the_arena():
    from phasegen.trainer import GradBuckets
    from phasegen.unet import BACKWARD_ORDER, ParamArena
    arena = ParamArena(16, torch.device("cpu"))
    b = GradBuckets(arena)
    assert list(b.spans) == BACKWARD_ORDER == ["U0", "U1", "U2", "U3", "D3", "D2", "D1", "D0"]
    assert b.covers_arena()
    # U0's bucket is the largest and is produced first (best case for overlap with the remaining backward GEMMs)
    sizes = {n: e - s for n, (s, e) in b.spans.items()}
    assert max(sizes, key=sizes.get) == "U0"
    # each bucket holds its conv weight and (where present) its BatchNorm affine pair
    s, e = b.spans["U1"]
    for key in (detgen.K_U1, detgen.BN_U1 + ".weight", detgen.BN_U1 + ".bias"):
        assert s <= arena.offsets[key] < e


def test_single_process_reducer_is_a_no_op():
    from phasegen.trainer import BucketedAllReduce
    from phasegen.unet import ParamArena
    arena = ParamArena(8, torch.device("cpu"))
    arena.grad.fill_(3.0)
    red = BucketedAllReduce(arena)
    assert red.world == 1
    red.launch("U0")
    assert red.wait_all() == ["U0"] and float(arena.grad.min()) == 3.0

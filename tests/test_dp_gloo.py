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


def worker(rank, port, q, compress=None, WORLD=WORLD):
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


@pytest.mark.parametrize("compress,WORLD", [(None, 2), ("bf16", 2), (None, 4)], ids=["fp32-2-ranks", "bf16-2-ranks", "fp32-4-ranks"])
def test_bucketed_allreduce_equals_mean_of_replica_grads(compress, WORLD):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, port, q, compress, WORLD)) for r in range(WORLD)]
    for p in procs:
        p.start()
    avg = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = [shard_grads(r) for r in range(WORLD)]
    for k in detgen.param_order():
        w = sum(wr[k] for wr in want).numpy() / WORLD
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


# ------------------------------------------------------------------------------------------------------------------
# The REAL Trainer.step control flow on two CPU ranks: engine.backward -> on_grads_ready -> BucketedAllReduce.launch (async,
# from inside backward, bucket by bucket) -> wait_all -> optim.step(grad_scale = 1/world), fed by the product's
# SpectrogramLoader shards.  Only the three device kernels' stand-ins differ from the GPU path: the engine's forward /
# backward, the loss and Adam are the oracle's (injected through Trainer's loss_fn / optim hooks).
# ------------------------------------------------------------------------------------------------------------------
N_CLIPS, B_LOADER, STEPS = 9, 2, 2          # 9 clips, world 2 x batch 2: 8 usable -> 2 full batches per rank and epoch


class OracleEngine:
    """CPU stand-in with UNetEngine's interface (arena, forward, backward(g_out, on_grads_ready), layer_param_keys)."""

    def __init__(self, Cc):
        from phasegen.unet import ParamArena
        self.C, self.device = Cc, torch.device("cpu")
        self.arena = ParamArena(Cc, self.device)
        self.calls = []

    def forward(self, x):
        from oracle import unet_ref
        self.p = {k: self.arena.p(k).clone().requires_grad_(True) for k in detgen.param_order()}
        self.out = unet_ref.unet_forward(self.p, x)          # BatchNorm statistics of THIS rank's shard only (model.py:40-41)
        return self.out.detach()

    def backward(self, g_out, on_grads_ready=None):
        from phasegen.unet import BACKWARD_ORDER, BN_OF, LAYERS
        self.out.backward(g_out)
        for name in BACKWARD_ORDER:                          # gradients become visible layer by layer, outermost up-conv first
            keys = [LAYERS[name][0]] + ([BN_OF[name] + ".weight", BN_OF[name] + ".bias"] if name in BN_OF else [])
            for k in keys:
                self.arena.g(k).copy_(self.p[k].grad)
            self.calls.append(name)
            if on_grads_ready is not None:
                on_grads_ready(name)


def oracle_loss(pred, batch, dpred, losses, mag_weight):
    from oracle import unet_ref
    pr = pred.clone().requires_grad_(True)
    loss, ang, mag = unet_ref.phase_loss(pr, batch)
    loss.backward()
    dpred.copy_(pr.grad)
    losses.copy_(torch.stack([loss.detach(), ang.detach(), mag.detach()]))
    return losses


class OracleAdam:
    def __init__(self, arena):
        self.arena, self.t = arena, 0
        self.m, self.v = torch.zeros_like(arena.flat), torch.zeros_like(arena.flat)

    def step(self, grad_scale=1.0):
        from oracle import unet_ref
        self.t += 1
        unet_ref.adam_step(self.arena.flat, self.arena.grad * grad_scale, self.m, self.v, self.t)


class StubModel:
    def __init__(self, Cc):
        self.engine = OracleEngine(Cc)
        self.engine.arena.load_numpy(detgen.make_params(Cc, seed=0))

    def parameters(self):
        return [self.engine.arena.p(k) for k in detgen.param_order()]


def dataset():
    return torch.from_numpy(detgen.make_batch(N_CLIPS, C, L, seed=7))


def trainer_worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from phasegen.data import SpectrogramLoader
        from phasegen.trainer import Trainer
        from phasegen.unet import BACKWARD_ORDER
        torch.set_num_threads(2)
        model = StubModel(C)
        tr = Trainer(model, loss_fn=oracle_loss, optim=OracleAdam(model.engine.arena))
        assert tr.world == WORLD
        data = dataset()
        loader = SpectrogramLoader(data, torch.zeros(N_CLIPS, 1), B_LOADER, True, rank, WORLD, seed=0)
        steps, losses = 0, []
        for d in loader:
            assert d[0].shape[0] == B_LOADER                 # no short batch on any rank
            losses.append(tr.step(d[0]).clone())
            assert not tr.reducer.pending and model.engine.calls[-8:] == BACKWARD_ORDER
            steps += 1
        q.put((rank, steps, model.engine.arena.flat.clone().numpy(), torch.stack(losses).numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_trainer_step_world2_equals_reference_semantics():
    """Two gloo ranks run Trainer.step on their loader shards; the parameters on BOTH ranks must equal a single-process
    restatement of nn.parallel.data_parallel's semantics: per-replica BatchNorm, gradient = mean over replicas of each
    replica's mean-loss gradient (model.py:40-41), one Adam update per step."""
    from oracle import unet_ref
    from phasegen.data import SpectrogramLoader
    from phasegen.unet import ParamArena
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=trainer_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(WORLD):
        r, steps, flat, losses = q.get(timeout=300)
        got[r] = (steps, flat, losses)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] == STEPS                              # same number of steps (= collectives) on every rank
    assert np.array_equal(got[0][1], got[1][1])                         # replicas stay bit-identical

    # single-process reference
    arena = ParamArena(C, torch.device("cpu"))
    arena.load_numpy(detgen.make_params(C, seed=0))
    opt = OracleAdam(arena)
    data = dataset()
    shards = [list(SpectrogramLoader(data, torch.zeros(N_CLIPS, 1), B_LOADER, True, r, WORLD, seed=0)) for r in range(WORLD)]
    seen = torch.cat([b[0] for sh in shards for b in sh])
    assert seen.shape[0] == 8 and len({t.numpy().tobytes() for t in seen}) == 8     # disjoint shards of one permutation
    for s in range(STEPS):
        gsum = torch.zeros_like(arena.grad)
        for r in range(WORLD):
            p = {k: arena.p(k).clone().requires_grad_(True) for k in detgen.param_order()}
            batch = shards[r][s][0]
            loss, ang, mag = unet_ref.phase_loss(unet_ref.unet_forward(p, batch[:, 0]), batch)
            loss.backward()
            tmp = ParamArena(C, torch.device("cpu"))
            for k in detgen.param_order():
                tmp.g(k).copy_(p[k].grad)
            gsum += tmp.grad
            assert abs(loss.item() - float(got[r][2][s][0])) <= 1e-6 * abs(loss.item())
        arena.grad.copy_(gsum)
        opt.step(grad_scale=1.0 / WORLD)
    want = arena.flat.numpy()
    assert np.max(np.abs(got[0][1] - want)) <= 1e-6 * np.max(np.abs(want))


@pytest.mark.parametrize("n,world,batch,steps", [(47, 2, 16, 1), (100, 2, 16, 3), (64, 4, 16, 1), (129, 8, 16, 1), (40, 1, 16, 2)])
def test_loader_gives_every_rank_the_same_number_of_full_batches(n, world, batch, steps):
    """perm[rank::world] alone gave ranks different step counts when n % (world * batch) != 0 (e.g. n = 31, W = 2,
    B = 16: rank 0 one step, rank 1 none), pairing collectives of different steps.  Every epoch is now cut to a whole number of
    global batches."""
    from phasegen.data import SpectrogramLoader
    data = torch.arange(n, dtype=torch.float32)[:, None]
    full, seen = [], []
    for r in range(world):
        ld = SpectrogramLoader(data, torch.zeros(n, 1), batch, True, r, world, seed=3)
        bs = [b[0] for b in ld]
        full.append(sum(1 for b in bs if b.shape[0] == batch))             # train.py:38-39 skips short batches
        if world > 1:
            assert all(b.shape[0] == batch for b in bs) and len(ld) == len(bs)
        seen += [int(v) for b in bs for v in b[:, 0]]
    assert full == [steps] * world
    assert len(seen) == len(set(seen))                                     # ranks never share a clip within an epoch


def test_loader_refuses_less_than_one_global_batch():
    from phasegen.data import SpectrogramLoader
    data = torch.zeros(31, 1)
    with pytest.raises(ValueError, match="global batch"):
        list(SpectrogramLoader(data, torch.zeros(31, 1), 16, True, 0, 2, seed=0))


def replica_worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from phasegen.trainer import sync_replicas
        from phasegen.unet import ParamArena
        arena = ParamArena(C, torch.device("cpu"))
        arena.load_numpy(detgen.make_params(C, seed=0))
        sync_replicas(arena, None, "check")                                  # identical replicas pass
        arena.p(detgen.param_order()[3]).view(-1)[5] += 1e-3 * rank          # rank 1 now differs in ONE element
        try:
            sync_replicas(arena, None, "check")
            raised = False
        except RuntimeError as e:
            raised = "replicas differ" in str(e)
        arena.buffers[detgen.BN_KEYS[0] + ".running_var"][2] += 0.5 * rank   # ... and in one BatchNorm buffer
        sync_replicas(arena, None, "broadcast")
        sync_replicas(arena, None, "check")                                  # rank 0's state everywhere
        q.put((rank, raised, arena.flat.clone().numpy(), arena.buffers[detgen.BN_KEYS[0] + ".running_var"].clone().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_trainer_refuses_diverged_replicas_and_can_broadcast():
    """default-initialised weights come from torch's global RNG, so ranks that were not seeded identically would
    train diverged replicas silently.  Trainer.__init__ compares a checksum across ranks (raises) or broadcasts rank 0's state."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ps = [ctx.Process(target=replica_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in ps:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(WORLD)), key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert all(g[1] for g in got)                       # every rank saw the mismatch
    assert np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][3], got[1][3])

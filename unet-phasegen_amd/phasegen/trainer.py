"""The training step of train.py:41-62 as one fused pipeline, plus multi-GPU data parallelism.

    optim.zero_grad()            -> nothing to do (gradient arena is overwritten)
    pred = model.forward(x)      -> engine.forward on batch[:, 0] read in place
    loss (cos/sin/mag MSE)       -> pg_loss_fwd_bwd (loss and d loss / d pred in one pass; targets' cos/sin are
                                    computed on device, replacing the 3 extra H2D copies of train.py:49-57)
    loss.backward()              -> engine.backward
    optim.step()                 -> one pg_adam_step over the arena

Data parallel (replaces nn.parallel.data_parallel, model.py:40-41): one process per GPU, RCCL through
torch.distributed.  Each layer's gradient bucket (conv weight + its BN affine, contiguous in the arena) is
all-reduced asynchronously as soon as backward has produced it -- U0's 1.07 GB bucket first, while the remaining
dgrad / wgrad GEMMs keep the matrix cores busy -- and the 1/world averaging is folded into Adam's grad_scale.
BatchNorm statistics stay per replica, exactly as data_parallel computes them (no SyncBN); running stats are
rank-local (rank 0's are the ones saved).
"""
import torch
import torch.distributed as dist

from . import ops
from .optim import Adam
from .unet import BACKWARD_ORDER, BN_OF, LAYERS


class GradBuckets:
    """Contiguous gradient-arena ranges, one per layer, in the order backward completes them (U0 ... D0)."""

    def __init__(self, arena):
        self.arena = arena
        self.spans = {}
        for name in BACKWARD_ORDER:
            keys = [LAYERS[name][0]]
            if name in BN_OF:
                keys += [BN_OF[name] + ".weight", BN_OF[name] + ".bias"]
            self.spans[name] = arena.span(keys)

    def view(self, name):
        s, e = self.spans[name]
        return self.arena.grad[s:e]

    def covers_arena(self):
        """True when the buckets tile the whole arena exactly once (no parameter missed, none reduced twice)."""
        ivs = sorted(self.spans.values())
        return ivs[0][0] == 0 and ivs[-1][1] == self.arena.numel and all(a[1] == b[0] for a, b in zip(ivs, ivs[1:]))


class BucketedAllReduce:
    """Asynchronous per-layer gradient all-reduce (sum) over torch.distributed -- RCCL over xGMI when the process
    group's backend is "nccl", gloo in the CPU tests.  ``launch(layer)`` is called from inside backward right after
    the layer's wgrad has been enqueued; ``wait_all()`` before the optimiser step.  The 1/world averaging is NOT
    applied here: Adam folds it into its single pass over the arena (grad_scale)."""

    def __init__(self, arena, group=None, always=False, compress=None):
        if compress not in (None, "bf16"):
            raise ValueError("compress: None (fp32 payload) or 'bf16'")
        self.buckets = GradBuckets(arena)
        self.group = group
        self.compress = compress     # "bf16": the payload on the wire is bf16 (half the xGMI bytes);
        self._wire = {}              #         the sum runs in bf16 inside the collective, the arena keeps fp32
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.always = always and dist.is_available() and dist.is_initialized()   # run the collective even at world 1 (tests)
        self.enabled = True          # False: launch() only records the order (bench.py's compute-only diagnostic step)
        self.pending = {}            # bucket name -> (work, wire, fp32 view), in launch order
        self.launched = []

    def launch(self, name):
        self.launched.append(name)
        if not self.enabled or not (self.world > 1 or self.always):
            return
        g = self.buckets.view(name)
        if self.compress is None:
            self.pending[name] = (dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None)
            return
        wire = self._wire.get(name)
        if wire is None:
            wire = self._wire[name] = torch.empty(g.shape, dtype=torch.bfloat16, device=g.device)
        wire.copy_(g)                                                          # fp32 -> bf16 (RNE) on the launch stream
        self.pending[name] = (dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True), wire, g)

    def wait(self, name):
        """Make the CURRENT stream wait for bucket ``name``'s collective (no-op if none is pending for it)."""
        item = self.pending.pop(name, None)
        if item is not None:
            w, wire, g = item
            w.wait()
            if wire is not None:
                g.copy_(wire)                                                  # bf16 sum -> fp32 arena

    def wait_all(self):
        for name in list(self.pending):
            self.wait(name)
        done, self.launched = self.launched, []
        return done


def arena_checksum(arena):
    """Two moments (sum, sum of squares; fp64) of the parameter arena followed by those of every floating BatchNorm buffer."""
    parts = [arena.flat] + [arena.buffers[k] for k in sorted(arena.buffers) if arena.buffers[k].is_floating_point()]
    chk = torch.zeros(2 * len(parts), device=arena.flat.device, dtype=torch.float64)
    for i, t in enumerate(parts):
        t = t.reshape(-1)
        for s0 in range(0, t.numel(), 1 << 26):          # chunked: no second fp64 copy of a 2.45 GB arena
            c = t[s0:s0 + (1 << 26)].double()
            chk[2 * i] += c.sum()
            chk[2 * i + 1] += (c * c).sum()
    return chk


def sync_replicas(arena, group=None, mode="check"):
    """See Trainer.__init__.  No-op without an initialised process group or at world size 1."""
    if mode not in ("check", "broadcast", None):
        raise ValueError("replicas: 'check', 'broadcast' or None")
    if mode is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return
    if mode == "broadcast":
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(arena.flat, src=src, group=group)
        for k in sorted(arena.buffers):
            dist.broadcast(arena.buffers[k], src=src, group=group)
        if hasattr(arena, "touch"):
            arena.touch()
        return
    chk = arena_checksum(arena)
    got = [torch.empty_like(chk) for _ in range(dist.get_world_size(group))]
    dist.all_gather(got, chk, group=group)
    bad = [r for r, t in enumerate(got) if not torch.equal(got[0], t)]
    if bad:
        raise RuntimeError(f"phasegen.Trainer: data-parallel replicas differ at construction (group ranks {bad} vs rank 0): seed every "
                           "rank identically before building the model (torch.manual_seed), load the same checkpoint everywhere, "
                           "or pass replicas='broadcast'")


class Trainer:
    """``loss_fn(pred, batch, dpred, losses, mag_weight)`` and ``optim`` default to the device kernels (pg_loss_fwd_bwd,
    fused Adam).  They are injection points for tests/test_dp_gloo.py, which drives THIS step's control flow (bucket launches
    from inside backward, wait_all, grad_scale = 1/world) on CPU ranks with an oracle-backed engine; the product never
    passes them."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, group=None, mag_weight=0.2, always_reduce=False,
                 grad_compress=None, loss_fn=None, optim=None, overlap_adam=True, fuse_adam=True, replicas="check"):
        self.model = model
        self.engine = model.engine
        # Data parallel assumes identical replicas at the start (data_parallel, model.py:40-41, re-broadcasts the parameters every
        # forward; here they are only ever changed by identical updates).  Default-initialised weights come from torch's global
        # RNG, so a caller who forgets torch.manual_seed on every rank would train diverged replicas silently:
        #   replicas="check"      (default) compare a checksum of parameters + BatchNorm buffers across ranks, raise on mismatch
        #   replicas="broadcast"  copy rank 0's parameters and buffers to every rank
        #   replicas=None         trust the caller
        sync_replicas(self.engine.arena, group, replicas)
        self.optim = optim if optim is not None else Adam(model.parameters(), lr=lr, betas=betas, eps=eps)
        self._loss = loss_fn if loss_fn is not None else ops.loss_fwd_bwd
        self.reducer = BucketedAllReduce(self.engine.arena, group, always=always_reduce, compress=grad_compress)
        self.world = self.reducer.world
        if self.world > 1 and hasattr(self.engine, "contended"):
            self.engine.contended = True        # RCCL's collective kernels run beside backward's convolutions
        self.mag_weight = mag_weight
        self.losses = torch.zeros(3, device=self.engine.device)
        self._dpred = {}
        # Adam is HBM-bound (28 B per parameter, 17 GB per step) while the convolutions of backward are MFMA-bound and leave
        # HBM almost idle: each layer's slice of the update runs on a SIDE stream as soon as backward no longer reads that
        # layer's parameters (and, data-parallel, as soon as its gradient bucket's all-reduce has completed), beside the
        # remaining backward kernels.  Elementwise, so bit-identical to one update over the whole arena after backward.
        self.overlap_adam = bool(overlap_adam and self.engine.device.type == "cuda" and hasattr(self.optim, "step_range"))
        self._side = torch.cuda.Stream(self.engine.device) if self.overlap_adam else None
        self._due = []
        # One GPU, no collective between gradient and update: the update of every conv weight (99.996 % of the parameters) runs
        # in the epilogue of its own wgrad kernel -- no separate pass over the arena (17 GB of HBM traffic per step become 14.7,
        # and they move while the other workgroups of the same kernel keep the MFMA pipes busy).  BatchNorm's gamma / beta (24 k
        # values) keep their small launches.  Bit-identical to the separate update.
        self.fuse_adam = bool(fuse_adam and self.overlap_adam and self.world == 1 and not self.reducer.always
                              and hasattr(self.optim, "fused_args"))

    def step(self, batch):
        """batch: (B, 2, C, L) = [logmag ; angle] on the device.  Returns the device tensor [loss, ang, mag]."""
        pred = self.engine.forward(batch[:, 0])
        dpred = self._dpred.get(pred.shape)
        if dpred is None:
            dpred = self._dpred[pred.shape] = torch.empty_like(pred)
        with ops.timed("hbm:loss", 12 * pred.numel()):           # per bin-frame: logmag, angle, two predictions read, two gradients written (24 B)
            self._loss(pred, batch, dpred, self.losses, self.mag_weight)
        if not self.overlap_adam:
            self.engine.backward(dpred, self.reducer.launch)
            self.reducer.wait_all()
            self.optim.step(grad_scale=1.0 / self.world)
            return self.losses
        self.optim.begin_step()
        if self.fuse_adam:
            self.engine.backward(dpred, self._bn_update, None, self.optim.fused_args)
            return self.losses
        self._due.clear()
        self.engine.backward(dpred, self._grads_ready, self._due.append)
        self._update_due()                                                       # the last layers: nothing left to hide under
        torch.cuda.current_stream(self.engine.device).wait_stream(self._side)    # the next forward reads the updated weights
        self.reducer.wait_all()                                                  # (nothing left pending: bookkeeping only)
        return self.losses

    def _bn_update(self, name):
        """Fused mode: layer ``name``'s conv weight is updated by its wgrad kernel; what is left is its BatchNorm's gamma / beta."""
        keys = self.engine.layer_param_keys(name)[1:]
        if keys:
            s, e = self.engine.arena.span(keys)
            self.optim.step_range(s, e, 1.0)

    def _grads_ready(self, name):
        """Called by backward right after layer ``name``'s wgrad has been enqueued: start its bucket's all-reduce, and start the
        update of the layers backward is already done with (their last reader, the dgrad, was enqueued earlier) -- HERE, so
        that the HBM-bound update runs beside the dgrad that follows, not beside a wgrad: measured on MI355X, the wgrad
        kernels lose as much time to a concurrent Adam as the overlap saves, the dgrad kernels lose none."""
        self.reducer.launch(name)
        self._update_due()

    def _update_due(self):
        if not self._due:
            return
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            for name in self._due:
                s, e = self.reducer.buckets.spans[name]
                self.reducer.wait(name)                      # data parallel: the bucket's all-reduce (RCCL's stream) is awaited HERE,
                with ops.timed("hbm:adam." + name, 28 * (e - s)):   # on the side stream, not on the stream that runs backward
                    self.optim.step_range(s, e, 1.0 / self.world)
        self._due.clear()

    # -- checkpoint / resume.  The reference saves the model only (model.py:45-48) and cannot
    # resume; the model file keeps that exact format (UNetModel.save) and the optimiser state goes next to it.
    def save_checkpoint(self, path):
        self.model.save(path)
        o = self.optim
        torch.save({"step": o.step_count, "lr": o.param_groups[0]["lr"], "betas": o.betas, "eps": o.eps,
                    "exp_avg": o.m.cpu(), "exp_avg_sq": o.v.cpu()}, path + ".optim")

    def load_checkpoint(self, path):
        self.model.load(path)
        sd = torch.load(path + ".optim", map_location="cpu", weights_only=True)
        o = self.optim
        o.step_count = int(sd["step"])
        o.param_groups[0]["lr"] = float(sd["lr"])
        o.m.copy_(sd["exp_avg"])
        o.v.copy_(sd["exp_avg_sq"])

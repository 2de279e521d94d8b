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
from .unet import BACKWARD_ORDER


class GradBuckets:
    """Contiguous arena ranges, one per layer, in the order backward completes them."""

    def __init__(self, engine):
        self.spans = {name: engine.arena.span(engine.layer_param_keys(name)) for name in BACKWARD_ORDER}

    def view(self, arena_grad, name):
        s, e = self.spans[name]
        return arena_grad[s:e]


class Trainer:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, group=None, mag_weight=0.2):
        self.model = model
        self.engine = model.engine
        self.optim = Adam(model.parameters(), lr=lr, betas=betas, eps=eps)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets = GradBuckets(self.engine)
        self.mag_weight = mag_weight
        self.losses = torch.zeros(3, device=self.engine.device)
        self._dpred = {}
        self._pending = []

    def _on_ready(self, name):
        if self.world > 1:
            self._pending.append(dist.all_reduce(self.buckets.view(self.engine.arena.grad, name), group=self.group, async_op=True))

    def step(self, batch):
        """batch: (B, 2, C, L) = [logmag ; angle] on the device.  Returns the device tensor [loss, ang, mag]."""
        pred = self.engine.forward(batch[:, 0])
        dpred = self._dpred.get(pred.shape)
        if dpred is None:
            dpred = self._dpred[pred.shape] = torch.empty_like(pred)
        ops.loss_fwd_bwd(pred, batch, dpred, self.losses, self.mag_weight)
        self.engine.backward(dpred, self._on_ready)
        for w in self._pending:
            w.wait()
        self._pending.clear()
        self.optim.step(grad_scale=1.0 / self.world)
        return self.losses

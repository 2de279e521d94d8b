"""Drop-in surface of the reference's ``model.py``: ``UNetModel(input_nc, output_nc, norm_layer, gpu_ids)`` with
``.forward / .save / .load / .parameters() / .cuda()`` and a ``.model`` attribute whose ``state_dict()`` carries the
reference's 38 key names (model.py:22-54), so checkpoints interchange and ``train.py`` / ``demo.py``
logic ports 1:1.  Underneath, every tensor lives in the engine's flat arenas and every op is a libphasegen kernel.
"""
import torch
import torch.nn as nn

from . import detgen
from .unet import UNetEngine


class _UNetFn(torch.autograd.Function):
    """Whole-network autograd node: lets the reference's ``loss.backward(); optim.step()`` loop run unchanged.

    The engine keeps ONE set of activations (the last forward's) and ONE gradient arena that backward overwrites, so two
    things the general autograd contract allows are refused loudly instead of giving silently wrong gradients: backward of
    an output that is not the engine's most recent forward, and gradient accumulation (backward while ``p.grad`` still
    holds the previous gradients -- the reference calls ``optim.zero_grad()`` every step, train.py:41)."""

    @staticmethod
    def forward(ctx, engine, x, *params):
        ctx.engine = engine
        ctx.params = params
        out = engine.forward(x.detach()).clone()
        ctx.fwd_id = engine.fwd_count
        return out

    @staticmethod
    def backward(ctx, g):
        eng = ctx.engine
        if eng.fwd_count != ctx.fwd_id:
            raise RuntimeError("phasegen UNetModel: backward through a forward that is no longer the engine's latest "
                               "(run no-grad forwards, e.g. validation, under torch.no_grad() AFTER loss.backward())")
        if any(p.grad is not None for p in ctx.params):
            raise RuntimeError("phasegen UNetModel: gradient accumulation is not supported -- call optim.zero_grad() "
                               "before loss.backward() (the gradient arena is overwritten by every backward)")
        eng.backward(g)
        return (None, None) + tuple(eng.arena.g(k) for k in detgen.param_order())


class _StateHolder(nn.Module):
    """Plays the role of ``UNetModel.model`` (the outermost UNetBlock): owns the parameters and speaks the
    reference's state-dict dialect (``model.0.weight`` ... ``model.4.num_batches_tracked``)."""

    def __init__(self, arena):
        super().__init__()
        self._arena = arena
        self._names = {}
        for k in detgen.param_order():
            par = nn.Parameter(arena.p(k))
            par._pg_arena, par._pg_key = arena, k
            safe = k.replace(".", "/")
            self.register_parameter(safe, par)
            self._names[k] = safe

    def param(self, k):
        return getattr(self, self._names[k])

    def state_dict(self, *args, **kwargs):
        out = {}
        for k in detgen.state_dict_order():
            out[k] = (self.param(k).detach() if k in self._names else self._arena.buffers[k])
        return out

    def load_state_dict(self, sd, strict=True):
        want = set(detgen.state_dict_order())
        missing = [k for k in want if k not in sd and not k.endswith("num_batches_tracked")]
        unexpected = [k for k in sd if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        self._arena.touch()
        with torch.no_grad():
            for k, v in sd.items():
                if k not in want:
                    continue
                dst = self.param(k) if k in self._names else self._arena.buffers[k]
                if tuple(dst.shape) != tuple(v.shape):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(dst.shape)}")
                dst.copy_(v)

    def cpu(self):            # model.py:46 calls self.model.cpu().state_dict(); the arena never leaves the device
        return _CpuView(self)

    def cuda(self, device=None):
        return self


class _CpuView:
    def __init__(self, holder):
        self._h = holder

    def state_dict(self):
        return {k: v.detach().cpu().clone() for k, v in self._h.state_dict().items()}


class UNetModel(nn.Module):
    def __init__(self, input_nc, output_nc, norm_layer=nn.BatchNorm2d, gpu_ids=[], precision=None):
        """Reference signature (model.py:23) + ``precision``: MFMA operand mode of this model's convolutions
        ("fp32" = the reference's arithmetic, "bf16x3", "bf16"; None = the calling thread's default, fp32)."""
        super().__init__()
        if output_nc != 2 * input_nc:
            raise NotImplementedError("phasegen UNetModel: output_nc must be 2*input_nc ([phase ; magnitude], train.py:45)")
        if norm_layer in (nn.InstanceNorm1d, nn.InstanceNorm2d):
            raise NotImplementedError("phasegen UNetModel: only BatchNorm (the reference's configuration) is implemented")
        self.gpu_ids = list(gpu_ids)
        dev = torch.device("cuda", self.gpu_ids[0]) if self.gpu_ids else None
        self.engine = UNetEngine(input_nc, dev, precision=precision)
        self.engine.arena.init_default()
        self.model = _StateHolder(self.engine.arena)

    # nn.parallel.data_parallel (model.py:40-41) is replaced by one process per GPU + RCCL (phasegen.trainer)
    def forward(self, input):
        if torch.is_tensor(input) and input.is_cuda and input.device != self.engine.device:
            input = input.to(self.engine.device)      # the reference moves its input to the model's GPU (.cuda(gpu_id))
        if torch.is_grad_enabled():
            params = [self.model.param(k) for k in detgen.param_order()]
            return _UNetFn.apply(self.engine, input, *params)
        return self.engine.forward(input, inference=True).clone()      # no-grad: nothing is kept for a backward

    def save(self, path):
        torch.save(self.model.cpu().state_dict(), path)

    def load(self, path):
        self.model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))

    def cuda(self, device=None):
        return self

    def load_numpy(self, params):
        self.engine.arena.load_numpy(params)
        return self

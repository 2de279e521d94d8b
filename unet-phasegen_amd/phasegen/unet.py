"""The U-Net phase generator of model.py on MI355X: parameter arena, activation plan, forward and backward.

Reference wiring (model.py:27-34, 85-113), with the names used throughout this repo:

    x0 --D0--> a0 --leaky,D1,BN--> h1 --leaky,D2,BN--> h2 --leaky,D3--> d3
    d3 --relu,U3,BN--> u3 ; cat2 = [h2 | u3] --relu,U2,BN--> u2 ; cat1 = [h1 | u2] --relu,U1,BN--> u1 ;
    cat0 = [a0 | u1] --relu,U0,BN--> out

MI355X-first layout decisions:
  * ONE flat fp32 arena for all 20 parameters (and a twin for gradients): Adam is a single streaming launch, the
    data-parallel all-reduce works on contiguous buckets of the same arena, a checkpoint is one copy.
  * torch.cat is never executed: each concat is one (B, 4C, L') buffer whose halves are written in place by the
    producing BatchNorm / conv (batch-stride-aware kernels).  The in-place (Leaky)ReLUs of model.py:80,82 are not
    separate passes either: PRODUCERS store pre-activated tensors -- a skip tensor h is written twice, as
    LeakyReLU(h) for the next down conv and as ReLU(h) (== ReLU(LeakyReLU(h))) into the concat buffer the up path
    reads -- so every conv of the network loads its operand with the identity (the fused activation-on-load of the
    C ABI costs ~6 % of MFMA issue slots in the matrix loop; measured).  BatchNorm backward uses the raw conv outputs,
    activation-derivative masks use the sign of the stored activated tensors (sign is preserved).
  * backward mirrors it: dgrad epilogues add the skip gradient and multiply by the activation derivative, wgrad
    overwrites the gradient arena (zero_grad folded in).
"""
import math

import numpy as np
import torch

from . import detgen, ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU

ALIGN = 64  # floats; keeps every parameter 256-B aligned inside the arena

# (key, kind, stride, pad): kind 'c' = Conv1d (Cout, Cin, k), 't' = ConvTranspose1d (Cin, Cout, k)
LAYERS = {
    "D0": (detgen.K_D0, "c", 2, 16), "D1": (detgen.K_D1, "c", 1, 2), "D2": (detgen.K_D2, "c", 2, 1),
    "D3": (detgen.K_D3, "c", 2, 1), "U3": (detgen.K_U3, "t", 2, 1), "U2": (detgen.K_U2, "t", 2, 1),
    "U1": (detgen.K_U1, "t", 1, 2), "U0": (detgen.K_U0, "t", 2, 16),
}
BN_OF = {"D1": detgen.BN_D1, "D2": detgen.BN_D2, "U3": detgen.BN_U3, "U2": detgen.BN_U2, "U1": detgen.BN_U1,
         "U0": detgen.BN_U0}
# order in which backward produces weight gradients (used for all-reduce bucketing)
BACKWARD_ORDER = ["U0", "U1", "U2", "U3", "D3", "D2", "D1", "D0"]

# Work split of backward's launches when a collective's kernels share the chip (engine.contended, set by the data-parallel Trainer):
# pg_conv_args.schedule bits OR-ed into the thread's default, per launch.  From profiles/r04_contention.json (tools/contention.py:
# every launch of backward at the headline shape beside a collective-shaped kernel -- 256 threads, <= 113 VGPRs, 32 KB LDS, streaming
# memory -- on 16 / 32 / 64 CUs): the fine stream-K split (PG_SCHED_CONTENDED) bounds the tail of the long F / T launches when CUs are
# slowed (dgrads of D1, D2, U0, U1, U2: -5 ... -14 % summed over the three hold sizes), the short k = 4 / k = 5 dgrads are as fast
# or faster on the automatic grids; the wgrads -- thousands of short tiles -- run ONE TILE PER WORKGROUP instead of several whole
# tiles per persistent workgroup: the hardware dispatcher then balances 8192 units instead of 2048 when some CUs have lost a slot
# (U0 wgrad paid a flat + 4 ms beside any hold; backward 117.4 against 120.5 - 121.7 ms at 16 / 32 / 64 held CUs, equal on a free
# chip); the two-waves-per-SIMD kernels (PG_SCHED_NO_RAW3) only win when the co-resident kernel is too fat to share a CU with
# conv_raw3 (194 VGPRs), which a collective is not.
CONTENDED_DGRAD = {"U0": ops.SCHED_CONTENDED, "U1": ops.SCHED_CONTENDED, "U2": ops.SCHED_CONTENDED, "U3": 0,
                   "D3": 0, "D2": ops.SCHED_CONTENDED, "D1": ops.SCHED_CONTENDED}
CONTENDED_WGRAD = {name: ops.SCHED_TILE_PER_WG for name in BACKWARD_ORDER}


def frame_plan(L):
    """Frame counts of every level; raises for lengths the U-Net cannot concatenate (valid: L % 8 == 0, L >= 24)."""
    L1 = ops.conv_out_len(L, 32, 2, 16)
    L2 = ops.conv_out_len(L1, 8, 1, 2)
    L3 = ops.conv_out_len(L2, 8, 2, 1)
    L4 = ops.conv_out_len(L3, 4, 2, 1)
    ok = (L4 >= 1 and ops.convt_out_len(L4, 5, 2, 1) == L3 and ops.convt_out_len(L3, 8, 2, 1) == L2
          and ops.convt_out_len(L2, 8, 1, 2) == L1 and ops.convt_out_len(L1, 32, 2, 16) == L)
    if not ok:
        raise ValueError(f"UNet: {L} frames cannot be skip-concatenated (need a multiple of 8, >= 24)")
    return L1, L2, L3, L4


class ParamArena:
    """All parameters in one flat device buffer, reference state-dict names, reference parameter order."""

    def __init__(self, C, device):
        self.C = C
        self.device = device
        shapes = detgen.conv_shapes(C)
        self.shapes, self.offsets = {}, {}
        off = 0
        for k in detgen.param_order():
            shp = shapes[k] if k in shapes else (2 * C,)
            self.shapes[k], self.offsets[k] = shp, off
            off += (int(np.prod(shp)) + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.version = 0             # bumped by everything that rewrites parameters (init, load, optimiser steps): the engine's
        #                              bf16 weight shadows are rebuilt when it changes; code that writes `flat` directly calls touch()
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.buffers = {}
        for k in detgen.BN_KEYS:
            self.buffers[k + ".running_mean"] = torch.zeros(2 * C, device=device)
            self.buffers[k + ".running_var"] = torch.ones(2 * C, device=device)
            self.buffers[k + ".num_batches_tracked"] = torch.zeros((), device=device, dtype=torch.long)

    def view(self, k, of=None):
        base = self.flat if of is None else of
        n = int(np.prod(self.shapes[k]))
        return base[self.offsets[k]: self.offsets[k] + n].view(self.shapes[k])

    def p(self, k):
        return self.view(k)

    def g(self, k):
        return self.view(k, self.grad)

    def span(self, keys):
        """(start, end) float offsets of the arena range covering ``keys`` (must be adjacent in the arena)."""
        s = min(self.offsets[k] for k in keys)
        e = max(self.offsets[k] + (int(np.prod(self.shapes[k])) + ALIGN - 1) // ALIGN * ALIGN for k in keys)
        return s, e

    def touch(self):
        self.version += 1

    def init_default(self, seed=None):
        """torch's default init (model.py builds stock nn.Conv1d / nn.ConvTranspose1d / BatchNorm; weights_init
        at model.py:12-20 is never called): conv weights U(+-1/sqrt(fan_in)), gamma 1, beta 0."""
        gen = torch.Generator(device=self.device)
        if seed is None:            # drawn FROM torch's global stream, as nn.Module init is: same manual_seed -> same weights, and a
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())     # second model built afterwards gets different ones
        gen.manual_seed(seed)
        for k, shp in self.shapes.items():
            v = self.view(k)
            if len(shp) == 3:
                b = 1.0 / math.sqrt(detgen.fan_in(k, shp))
                v.uniform_(-b, b, generator=gen)
            elif k.endswith(".weight"):
                v.fill_(1.0)
            else:
                v.zero_()
        self.touch()

    def load_numpy(self, params):
        self.touch()
        for k in self.shapes:
            self.view(k).copy_(torch.from_numpy(np.ascontiguousarray(params[k])))
        for k in self.buffers:
            if k in params:
                self.buffers[k].copy_(torch.as_tensor(np.asarray(params[k])))


class UNetEngine:
    """Forward / backward of the whole network through libphasegen, with a per-(B, L) activation plan."""

    def __init__(self, C, device=None, precision=None):
        if not torch.cuda.is_available():
            raise RuntimeError("phasegen.UNetEngine needs an MI355X (no CPU fallback exists for the hot path)")
        self.C = C
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        # MFMA operand mode of every conv of this engine (pg_conv_args.precision); None = the calling thread's default
        self.precision = None if precision is None else ops.precision_code(precision)
        self.graphs = False          # True: the bf16-resident inference forward replays a captured HIP graph per (batch, frames)
        self.contended = False       # True (set by the data-parallel Trainer): RCCL's kernels share the chip during backward ->
        #                              backward convs keep the fine stream-K split (pg_conv_args.schedule, PG_SCHED_CONTENDED)
        self.fwd_count = 0           # forward passes so far: backward() refers to the LAST one (checked by the autograd node)
        self.arena = ParamArena(C, self.device)
        self.plans = {}
        self.bn_save = {k: (torch.empty(2 * C, device=self.device), torch.empty(2 * C, device=self.device))
                        for k in BN_OF}
        self.cur = None

    # -- activation plan ---------------------------------------------------------------------------------------
    def plan(self, B, L):
        key = (B, L)
        if key not in self.plans:
            C = self.C
            L1, L2, L3, L4 = frame_plan(L)
            dev = self.device

            def z(*s):
                return torch.empty(*s, device=dev, dtype=torch.float32)
            # cat*: [relu(skip) | relu(up)]; l0/l1/l2: leaky(a0/h1/h2); d3: relu(d3); c*/r*: raw conv outputs (for BN bwd)
            f = dict(cat0=z(B, 4 * C, L1), cat1=z(B, 4 * C, L2), cat2=z(B, 4 * C, L3), d3=z(B, 4 * C, L4),
                     l0=z(B, 2 * C, L1), l1=z(B, 2 * C, L2), l2=z(B, 2 * C, L3),
                     c1=z(B, 2 * C, L2), c2=z(B, 2 * C, L3), r3=z(B, 2 * C, L3), r2=z(B, 2 * C, L2),
                     r1=z(B, 2 * C, L1), r0=z(B, 2 * C, L), out=z(B, 2 * C, L))
            self.plans[key] = dict(fwd=f, bwd=None, L=(L, L1, L2, L3, L4))
        return self.plans[key]

    def _bwd_bufs(self, plan, B):
        if plan["bwd"] is None:
            C = self.C
            L, L1, L2, L3, L4 = plan["L"]
            dev = self.device

            def z(*s):
                return torch.empty(*s, device=dev, dtype=torch.float32)
            plan["bwd"] = dict(g_r0=z(B, 2 * C, L), g_cat0=z(B, 4 * C, L1), g_r1=z(B, 2 * C, L1),
                               g_cat1=z(B, 4 * C, L2), g_r2=z(B, 2 * C, L2), g_cat2=z(B, 4 * C, L3),
                               g_r3=z(B, 2 * C, L3), g_d3=z(B, 4 * C, L4), g_c2=z(B, 2 * C, L3), g_c1=z(B, 2 * C, L2))
        return plan["bwd"]

    # -- helpers -----------------------------------------------------------------------------------------------
    def _conv(self, name, x, y, y_act=ACT_NONE, y2=None, y2_act=ACT_NONE):
        key, kind, s, p = LAYERS[name]
        with ops.timed(name + ".fwd"):
            ops.conv_fwd(x, self.arena.p(key), y, s, p, transposed=(kind == "t"), y_act=y_act, y2=y2, y2_act=y2_act,
                         precision=self.precision)

    def _bn(self, name, x, y, update_stats, y_act=ACT_NONE, y2=None, y2_act=ACT_NONE):
        key = BN_OF[name]
        a = self.arena
        sm, si = self.bn_save[name]
        rm = a.buffers[key + ".running_mean"] if update_stats else None
        rv = a.buffers[key + ".running_var"] if update_stats else None
        nb = a.buffers[key + ".num_batches_tracked"] if update_stats else None
        with ops.timed("hbm:bn_fwd." + name, 4 * x.numel() * (2 + (y2 is not None))):        # one read, one or two fp32 writes
            ops.bn_fwd(x, y, a.p(key + ".weight"), a.p(key + ".bias"), sm, si, rm, rv, y_act=y_act, y2=y2, y2_act=y2_act,
                       num_batches_tracked=nb)

    # -- bf16-resident inference forward (BASELINE configs[4]) ----------------------------------------------------------------
    RESIDENT_LAYERS = ("D0", "D1", "D2", "D3", "U3", "U2", "U1", "U0")  # U3 (k = 5): shadow padded to 4 taps per phase

    def resident_ok(self, B=None, L=None):
        """bf16-resident forward: precision bf16, channel counts the 32-deep slabs divide (C % 8 == 0) and -- for a given
        batch / frame count -- every layer's windows fit the kernels' slots (many very short samples per tile do not)."""
        p = self.precision if self.precision is not None else ops._tls.precision
        if p != 1 or self.C % 8:
            return False
        if B is None:
            return True
        key = ("hok", B, L)
        if key not in self.plans:
            Ls = (L,) + frame_plan(L)
            lin = {"D0": Ls[0], "D1": Ls[1], "D2": Ls[2], "D3": Ls[3], "U3": Ls[4], "U2": Ls[3], "U1": Ls[2], "U0": Ls[1]}
            self.plans[key] = all(ops.conv_fwd_h_supported(B, self.arena.shapes[LAYERS[n][0]], lin[n], LAYERS[n][2], LAYERS[n][3],
                                                           LAYERS[n][1] == "t") for n in self.RESIDENT_LAYERS)
        return self.plans[key]

    def _shadows(self):
        """bf16 shadows of the conv weights in the resident kernels' layout, rebuilt when the parameters have changed."""
        # two counters: arena.version is bumped by the writers that go through the ctypes kernels (init, load, phasegen's Adam:
        # torch cannot see those writes), arena.flat._version by every in-place write torch itself makes through a parameter
        # view (torch.optim.Adam(model.parameters()).step(), p.copy_/add_ under no_grad, p.detach().mul_, ...) -- either one changing
        # rebuilds the shadows.  Only a write through `p.data` (a tensor with its own version counter) still needs arena.touch().
        ver = (self.arena.version, self.arena.flat._version)
        if getattr(self, "_shadow_version", None) != ver:
            self._shadow = getattr(self, "_shadow", {})
            for name in self.RESIDENT_LAYERS:
                key, kind, s, p = LAYERS[name]
                self._shadow[name] = ops.shadow_weights(self.arena.p(key), kind == "t", s, out=self._shadow.get(name))
            self._shadow_version = ver
        return self._shadow

    def _plan_h(self, B, L):
        key = ("h", B, L)
        if key not in self.plans:
            C = self.C
            L1, L2, L3, L4 = frame_plan(L)
            dev = self.device

            def z(*s):
                return torch.empty(*s, device=dev, dtype=torch.float32)

            def zh(ch, frames):
                return ops.h_alloc(B, ch, frames, dev)       # zero-filled once: producers never write the row tails
            f = dict(x0=zh(C, L), l0=zh(2 * C, L1), l1=zh(2 * C, L2), l2=zh(2 * C, L3),
                     cat0=zh(4 * C, L1), cat1=zh(4 * C, L2), cat2=zh(4 * C, L3),
                     d3=zh(4 * C, L4), c1=z(B, 2 * C, L2), c2=z(B, 2 * C, L3), r3=z(B, 2 * C, L3), r2=z(B, 2 * C, L2),
                     r1=z(B, 2 * C, L1), r0=z(B, 2 * C, L), out=z(B, 2 * C, L))
            self.plans[key] = dict(fwd=f, L=(L, L1, L2, L3, L4))
        return self.plans[key]

    def _forward_resident(self, x, update_stats):
        """Forward with bf16-resident operands: every activation a conv reads lives in HBM as bf16 (written activated by the
        producing conv epilogue / BatchNorm), weights come from bf16 shadows, BatchNorm statistics and outputs stay fp32.
        No tensors are kept for backward: inference only.  With ``self.graphs`` the launch sequence behind the input cast (8 convs
        with their fixups + 6 BatchNorms) is captured once per (batch, frames, update_stats) into a HIP graph and replayed: every
        buffer it touches is a plan buffer, the weight shadows included (rebuilt IN PLACE when the parameters change)."""
        B, C, L = x.shape
        plan = self._plan_h(B, L)
        f = plan["fwd"]
        sh = self._shadows()
        ops.cast_rows_bf16(x, f["x0"])
        if not self.graphs or ops._timer is not None:
            self._resident_body(plan, sh, update_stats)
        else:
            # everything a capture freezes is part of the key: shape, the BatchNorm-buffer update, the thread's schedule word and the
            # engine's precision (ops reads both at launch time; a replay would silently keep the captured ones)
            key = ("graph", B, L, bool(update_stats), ops.current_schedule(), self.precision)
            g = self.plans.get(key)
            if g is None:                        # first forward at this shape: eager (first-launch costs stay out of the capture)
                self._resident_body(plan, sh, update_stats)
                self.plans[key] = "warm"
            else:
                if g == "warm":                  # second: capture -- nothing executes while capturing -- then replay as this call's forward
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self._resident_body(plan, sh, update_stats)
                        # ops' scratch caches are keyed by (device, CURRENT stream) and torch captures on a side stream of its own:
                        # the stream-K workspace the captured launches point at was allocated just now, for the capture stream.  It
                        # is held HERE, next to the graph, so that neither the caches' LRU eviction nor ops.release_workspaces()
                        # can free memory a live graph still references.
                        held = ops.conv_workspace(self.device)
                    g = self.plans[key] = (graph, held)
                g[0].replay()
        self.fwd_count += 1
        self.cur = None                      # nothing kept for backward
        return f["out"]

    def _resident_body(self, plan, sh, update_stats):
        f = plan["fwd"]
        L, L1, L2, L3, L4 = plan["L"]
        C = self.C
        h = 2 * C
        a = self.arena

        def conv(name, xh, Lin, **out):
            key, kind, s, p = LAYERS[name]
            with ops.timed(name + ".fwd"):
                ops.conv_fwd_h(xh, Lin, sh[name], a.shapes[key], s, p, transposed=(kind == "t"), **out)

        def bn(name, raw, **out):
            key = BN_OF[name]
            sm, si = self.bn_save[name]
            rm = a.buffers[key + ".running_mean"] if update_stats else None
            rv = a.buffers[key + ".running_var"] if update_stats else None
            nb = a.buffers[key + ".num_batches_tracked"] if update_stats else None
            ops.bn_fwd(raw, out.pop("y", None), a.p(key + ".weight"), a.p(key + ".bias"), sm, si, rm, rv, num_batches_tracked=nb, **out)

        conv("D0", f["x0"], L, yh=f["l0"], yh_act=ACT_LEAKY, yh2=f["cat0"][:, :h], yh2_act=ACT_RELU)
        conv("D1", f["l0"], L1, y=f["c1"])
        bn("D1", f["c1"], yh=f["l1"], yh_act=ACT_LEAKY, yh2=f["cat1"][:, :h], yh2_act=ACT_RELU)
        conv("D2", f["l1"], L2, y=f["c2"])
        bn("D2", f["c2"], yh=f["l2"], yh_act=ACT_LEAKY, yh2=f["cat2"][:, :h], yh2_act=ACT_RELU)
        conv("D3", f["l2"], L3, yh=f["d3"], yh_act=ACT_RELU)
        conv("U3", f["d3"], L4, y=f["r3"])
        bn("U3", f["r3"], yh=f["cat2"][:, h:], yh_act=ACT_RELU)
        conv("U2", f["cat2"], L3, y=f["r2"])
        bn("U2", f["r2"], yh=f["cat1"][:, h:], yh_act=ACT_RELU)
        conv("U1", f["cat1"], L2, y=f["r1"])
        bn("U1", f["r1"], yh=f["cat0"][:, h:], yh_act=ACT_RELU)
        conv("U0", f["cat0"], L1, y=f["r0"])
        bn("U0", f["r0"], y=f["out"])

    # -- forward -----------------------------------------------------------------------------------------------
    def forward(self, x, update_stats=True, inference=False):
        """x: (B, C, L) fp32 device tensor -> (B, 2C, L).  BatchNorm is ALWAYS in training mode, as in the
        reference (no .eval() anywhere; demo.py:36 runs batch-of-1 statistics).  ``inference=True`` promises that no
        backward follows: with precision bf16 the forward then runs on the bf16-resident kernels (csrc/conv_h3.hip)."""
        if x.dim() != 3 or x.shape[1] != self.C:
            raise ValueError(f"UNet: expected input (B, {self.C}, L), got {tuple(x.shape)}")
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError("UNet: input must be a float32 device tensor")
        if x.device != self.device:
            raise ValueError(f"UNet: input is on {x.device}, the engine on {self.device}")
        if x.stride(2) != 1 or (x.shape[1] > 1 and x.stride(1) != x.shape[2]):
            x = x.contiguous()                   # any other layout than (batch-strided) rows of contiguous frames: one copy
        with torch.cuda.device(self.device):     # kernels launch on the CURRENT device's stream: make that the engine's
            if inference and self.resident_ok(x.shape[0], x.shape[2]):
                return self._forward_resident(x, update_stats)
            return self._forward(x, update_stats)

    def _forward(self, x, update_stats):
        B, C, L = x.shape           # batch-strided views (e.g. batch[:, 0] of a (B,2,C,L) batch) are read in place
        plan = self.plan(B, L)
        f = plan["fwd"]
        h = 2 * C
        self._conv("D0", x, f["l0"], ACT_LEAKY, f["cat0"][:, :h], ACT_RELU)         # a0 -> leaky(a0), relu(a0)
        self._conv("D1", f["l0"], f["c1"])
        self._bn("D1", f["c1"], f["l1"], update_stats, ACT_LEAKY, f["cat1"][:, :h], ACT_RELU)   # h1
        self._conv("D2", f["l1"], f["c2"])
        self._bn("D2", f["c2"], f["l2"], update_stats, ACT_LEAKY, f["cat2"][:, :h], ACT_RELU)   # h2
        self._conv("D3", f["l2"], f["d3"], ACT_RELU)                                  # relu(d3)
        self._conv("U3", f["d3"], f["r3"])
        self._bn("U3", f["r3"], f["cat2"][:, h:], update_stats, ACT_RELU)             # relu(u3)
        self._conv("U2", f["cat2"], f["r2"])
        self._bn("U2", f["r2"], f["cat1"][:, h:], update_stats, ACT_RELU)
        self._conv("U1", f["cat1"], f["r1"])
        self._bn("U1", f["r1"], f["cat0"][:, h:], update_stats, ACT_RELU)
        self._conv("U0", f["cat0"], f["r0"])
        self._bn("U0", f["r0"], f["out"], update_stats)
        self.fwd_count += 1
        self.cur = (plan, x)
        return f["out"]

    # -- backward ----------------------------------------------------------------------------------------------
    def backward(self, g_out, on_grads_ready=None, on_layer_done=None, fused_adam=None):
        """Gradients of all 20 parameters into the gradient arena (overwritten).  ``on_grads_ready(layer)`` is
        called right after the kernels that complete a layer's gradients (conv weight + its BatchNorm's gamma/beta)
        have been enqueued -- the data-parallel wrapper launches that bucket's all-reduce from it.  ``on_layer_done(layer)``
        is called once backward has enqueued its LAST kernel that reads the layer's parameters (the dgrad): from there on
        the optimiser may overwrite them while the remaining layers' backward kernels run.  ``fused_adam(key)`` (optional)
        returns the pg_adam_args of conv weight ``key``: its update then runs in the epilogue of that weight's wgrad kernel, which
        is ordered after the layer's dgrad (single-GPU training; gradients are still written)."""
        if self.cur is None:
            raise RuntimeError("UNet.backward called before forward")
        with torch.cuda.device(self.device):
            self._backward(g_out, on_grads_ready, on_layer_done, fused_adam)

    def _backward(self, g_out, on_grads_ready, on_layer_done=None, fused_adam=None):
        plan, x0 = self.cur
        f = plan["fwd"]
        B = x0.shape[0]
        g = self._bwd_bufs(plan, B)
        a = self.arena
        h = 2 * self.C
        g_out = g_out.contiguous()
        def sched(table, name):          # per-launch schedule word of a data-parallel backward (None: the thread's default)
            return (ops.current_schedule() | table[name]) if self.contended else None

        def bn_bwd(name, raw, dy, dx):
            key = BN_OF[name]
            sm, si = self.bn_save[name]
            with ops.timed("hbm:bn_bwd." + name, 12 * raw.numel()):                         # x and dy read, dx written
                ops.bn_bwd(raw, dy, dx, a.p(key + ".weight"), sm, si, a.g(key + ".weight"), a.g(key + ".bias"))

        def wgrad(name, x, dy, act):
            key, kind, s, p = LAYERS[name]
            adam = fused_adam(key) if fused_adam is not None else None       # kept alive until the call has returned
            with ops.timed(name + ".wgrad"):
                ops.conv_wgrad(x, dy, a.g(key), s, p, x_act=act, transposed=(kind == "t"), precision=self.precision,
                               schedule=sched(CONTENDED_WGRAD, name), adam=adam)

        def dgrad(name, dy, dx, **kw):
            key, kind, s, p = LAYERS[name]
            with ops.timed(name + ".dgrad"):
                ops.conv_dgrad(dy, a.p(key), dx, s, p, transposed=(kind == "t"), precision=self.precision,
                               schedule=sched(CONTENDED_DGRAD, name), **kw)

        def ready(name):
            if on_grads_ready is not None:
                on_grads_ready(name)

        def done(name):
            if on_layer_done is not None:
                on_layer_done(name)

        def layer(name, wg, dg=None, **dkw):
            """One layer's two gradient kernels.  Plain: wgrad first, so that a data-parallel caller can start the bucket's
            all-reduce (``ready``) under the dgrad.  With the optimiser step fused into the wgrad epilogue the dgrad -- the last
            reader of the old weight -- must come first."""
            if fused_adam is None:
                wgrad(name, *wg)
                ready(name)
                if dg is not None:
                    dgrad(name, *dg, **dkw)
            else:
                if dg is not None:
                    dgrad(name, *dg, **dkw)
                wgrad(name, *wg)
                ready(name)
            done(name)

        # up path, outermost first.  Operands are the stored activated tensors (identity on load); the mask relu'(.) is
        # taken from their sign.
        for name, cat, raw, gin, g_raw, g_cat in (("U0", "cat0", "r0", g_out, "g_r0", "g_cat0"),
                                                  ("U1", "cat1", "r1", g["g_cat0"][:, h:], "g_r1", "g_cat1"),
                                                  ("U2", "cat2", "r2", g["g_cat1"][:, h:], "g_r2", "g_cat2"),
                                                  ("U3", "d3", "r3", g["g_cat2"][:, h:], "g_r3", "g_d3")):
            bn_bwd(name, f[raw], gin, g[g_raw])
            layer(name, (f[cat], g[g_raw], ACT_NONE), (g[g_raw], g[g_cat]), ref=f[cat], mask=ACT_RELU)
        # down path, innermost first; each dgrad adds the skip gradient and applies leaky' (sign of the stored leaky(h))
        layer("D3", (f["l2"], g["g_d3"], ACT_NONE), (g["g_d3"], g["g_cat2"][:, :h]), add=g["g_cat2"][:, :h], ref=f["l2"], mask=ACT_LEAKY)
        bn_bwd("D2", f["c2"], g["g_cat2"][:, :h], g["g_c2"])
        layer("D2", (f["l1"], g["g_c2"], ACT_NONE), (g["g_c2"], g["g_cat1"][:, :h]), add=g["g_cat1"][:, :h], ref=f["l1"], mask=ACT_LEAKY)
        bn_bwd("D1", f["c1"], g["g_cat1"][:, :h], g["g_c1"])
        layer("D1", (f["l0"], g["g_c1"], ACT_NONE), (g["g_c1"], g["g_cat0"][:, :h]), add=g["g_cat0"][:, :h], ref=f["l0"], mask=ACT_LEAKY)
        layer("D0", (x0, g["g_cat0"][:, :h], ACT_NONE))      # network input needs no dgrad

    def layer_param_keys(self, name):
        keys = [LAYERS[name][0]]
        if name in BN_OF:
            keys += [BN_OF[name] + ".weight", BN_OF[name] + ".bias"]
        return keys

    def intermediates(self):
        """Forward tensors by oracle name (test hook).  Raw conv outputs as they are; the tensors the reference would
        hold pre-activation are stored activated here: ``leaky:<name>`` / ``relu:<name>``."""
        plan, _ = self.cur
        f = plan["fwd"]
        h = 2 * self.C
        return {"leaky:a0": f["l0"], "relu:a0": f["cat0"][:, :h], "c1": f["c1"], "leaky:h1": f["l1"], "relu:h1": f["cat1"][:, :h],
                "c2": f["c2"], "leaky:h2": f["l2"], "relu:h2": f["cat2"][:, :h], "relu:d3": f["d3"], "r3": f["r3"],
                "relu:u3": f["cat2"][:, h:], "r2": f["r2"], "relu:u2": f["cat1"][:, h:], "r1": f["r1"],
                "relu:u1": f["cat0"][:, h:], "r0": f["r0"], "out": f["out"]}

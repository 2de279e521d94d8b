#!/usr/bin/env python3
"""bench.py -- spectrogram-frames/sec of the full training step (train.py:41-62: fwd + loss + bwd + Adam
[+ RCCL gradient all-reduce for N > 1]) of the C=1024 U-Net on synthetic (64, 2, 1024, 256) batches per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  How the numbers are taken:

  value / ms_per_step   K steps timed CLEAN (no events inside the region), barrier + synchronize on both sides, max over ranks.
  kernels               a SECOND pass (3 steps) with one HIP-event pair per conv launch on the launch stream; every label carries
                        the kernel symbol the library launches for it (pg_conv_describe: the name rocprofv3 reports).
  roofline              the kernel with the LARGEST total time per step in that pass (not the fastest one): its algorithmic FLOPs
                        per step / its time per step against the MFMA peak of the operand precision (fp32: 157.3 TFLOP/s,
                        MI355X_MICROARCH.md), plus `step_frac` (all 23 conv passes' FLOPs / the clean step time), `worst_kernel`
                        (lowest fraction among kernels holding >= 0.3 % of the step) and `by_kernel`.  Every entry can be
                        recomputed from profiles/rNN_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this same command):
                        calls and total time per kernel symbol there = launches_per_step x steps and ms_per_step here.
  cpu_baseline          the oracle (CPU restatement of the reference, oracle/unet_ref.py) doing the SAME training step on the
                        host cores on a bounded sample (batch 16 of the same C=1024, L=256 model); rank 0, N = 1 only.
  other_precisions      (N = 1, fp32 run only) the same step re-timed in the two optional MFMA operand modes ("bf16x3", "bf16").
                        Reported beside the headline, never as `value`.
  dp                    (N > 1) self-diagnosis of the data-parallel path: ranks that really took part in an all-reduce, a checksum
                        of the parameter arena compared across ranks (a diverged replica FAILS the bench), each bucket's
                        all-reduce timed alone, the compute-only step and the fraction of communication hidden under backward.

  other_configs         (N = 1, fp32 headline run only) the other BASELINE configurations measured in the SAME process, so that the
                        driver's record carries them too -- never as `value`:
                          fwd            configs[1]: forward only, batch 32 x 1024 x 256, fp32 (+ encoder_convs = D0-D3 together)
                          e2e            configs[4]: 32 stereo clips -> STFT 2048/512 + polar -> U-Net forward on the bf16-resident
                                         kernels -> ISTFT, with its own roofline against the dense bf16 MFMA peak
                          ref_default    the reference's own defaults (train.py:14-15): full step at batch 16 x 1024 x 128
                          dp_equivalent  the headline step with the Adam update NOT fused into the wgrad epilogues (per-layer
                                         slices on a side stream): what every rank of an N > 1 run executes, i.e. the N = 1 figure
                                         a scaling efficiency should divide by
                        Each can also be the whole line: --config fwd | e2e (same code, same numbers).
"""
import argparse
import json
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd"))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense bf16: 16 x the fp32 rate (v_mfma_f32_32x32x16_bf16: 32 cycles for 32 768 FLOP)
# What the chip SUSTAINS on that instruction with no memory traffic at all (tools/mfma_peak.py, profiles/r03_mfma_peak.txt: a
# register-only loop, every SIMD busy): 2460 TFLOP/s at 2.39 GHz on constant operands, 1778 TFLOP/s at 1.78 GHz on uniform random
# operands -- the power management clocks the MFMA-dense loop down with the bits that toggle.  `roofline.peak` stays the data-sheet
# figure; the bf16 lines also carry the fraction of this measured, power-limited rate.
SUSTAINED_BF16_MFMA_TFLOPS_RANDOM = 1778.0
PEAK_HBM_TBPS = 8.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s (about 6.3 TB/s is what a streaming kernel reaches)
DTYPES = {"fp32": "f32", "bf16": "bf16 operands / f32 accumulate",
          "bf16x3": "f32 operands split hi+lo into 3 bf16 MFMA products / f32 accumulate"}


def conv_flops(C, L, B):
    """Algorithmic FLOPs (2 x MAC, padding taps counted -- SURVEY.md §8d) per launch of every conv pass."""
    from phasegen.unet import frame_plan
    L1, L2, L3, L4 = frame_plan(L)
    # name: (Cin, Cout, k, positions the taps are applied at)  conv: output frames; convT: input frames
    g = {"D0": (C, 2 * C, 32, L1), "D1": (2 * C, 2 * C, 8, L2), "D2": (2 * C, 2 * C, 8, L3), "D3": (2 * C, 4 * C, 4, L4),
         "U3": (4 * C, 2 * C, 5, L4), "U2": (4 * C, 2 * C, 8, L3), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
    return {n: 2.0 * B * pos * ci * co * k for n, (ci, co, k, pos) in g.items()}


def device_info(torch, probe=True):
    """Which box and which clocks a line was measured on (a 13 % spread between boxes on the power-limited bf16 pipe went
    unexplained).  pci_bus / uuid identify the card; `mfma_probe` runs tools/dbg/mfma_peak.hip's register-only MFMA loops (no LDS, no
    memory, every SIMD) for a few ms and reports what THIS chip sustains right now: TFLOP/s and the shader clock (s_memtime ticks per
    s_memrealtime tick) for fp32 MFMA and for bf16 MFMA on random operands -- the power-limited ceiling of the bf16 path."""
    import ctypes
    p = torch.cuda.get_device_properties(torch.cuda.current_device())
    out = {"name": p.name, "arch": getattr(p, "gcnArchName", None), "cus": p.multi_processor_count,
           "pci_bus": "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0)),
           "uuid": str(getattr(p, "uuid", "")), "host": os.uname().nodename}
    so = os.path.join(ROOT, "tools", "dbg", "libmfma_peak.so")
    if probe and os.path.exists(so):
        try:
            lib = ctypes.CDLL(so)
            sig = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)] * 2
            lib.mfma_peak.argtypes = lib.mfma_peak_f32.argtypes = sig
            pr = {}
            for name, fn, iters in (("bf16_random_operands", lib.mfma_peak, 100000), ("fp32_random_operands", lib.mfma_peak_f32, 50000)):
                tf, mhz = ctypes.c_double(), ctypes.c_double()
                rc = fn(2 * p.multi_processor_count, 256, iters, 3, 1, ctypes.byref(tf), ctypes.byref(mhz))
                pr[name] = {"tflops": round(tf.value, 1), "shader_mhz": round(mhz.value), "rc": rc}
            out["mfma_probe"] = pr
        except Exception as e:          # noqa: BLE001
            out["mfma_probe"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def conv_bytes(C, L, B, fused_adam=True):
    """Algorithmic HBM bytes per launch of every conv pass (each operand once, fp32): fwd = x + w + y (+ the second activated copy
    where the engine stores one), dgrad = dy + w + dx (+ skip gradient and mask tensor read), wgrad = x + dy + dw (+ 24 B per
    parameter when the Adam update runs in its epilogue)."""
    from phasegen.unet import frame_plan
    L1, L2, L3, L4 = frame_plan(L)
    g = {"D0": (C, 2 * C, 32, L, L1), "D1": (2 * C, 2 * C, 8, L1, L2), "D2": (2 * C, 2 * C, 8, L2, L3), "D3": (2 * C, 4 * C, 4, L3, L4),
         "U3": (4 * C, 2 * C, 5, L4, L3), "U2": (4 * C, 2 * C, 8, L3, L2), "U1": (4 * C, 2 * C, 8, L2, L1), "U0": (4 * C, 2 * C, 32, L1, L)}
    two_copies = {"D0"}                                   # D0 stores leaky(a0) and relu(a0); the others feed a BatchNorm
    extra_reads = {"D3": 2, "D2": 2, "D1": 2, "U0": 1, "U1": 1, "U2": 1, "U3": 1}   # dgrad epilogue: skip gradient + mask tensor / mask tensor
    out = {}
    for n, (ci, co, k, li, lo) in g.items():
        w, x, y = 4 * ci * co * k, 4 * B * ci * li, 4 * B * co * lo
        out[n + ".fwd"] = w + x + y * (2 if n in two_copies else 1)
        out[n + ".dgrad"] = w + y + x * (1 + extra_reads.get(n, 0))
        out[n + ".wgrad"] = x + y + w + (6 * w if fused_adam else 0)
    return out


def host_threads():
    """Threads for the CPU leg: the affinity mask / cgroup quota, capped at the GPU box's per-GPU CPU share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/rNN_pmc_summary.json: FETCH_SIZE x2
    per the gfx950 correction + WRITE_SIZE, separate --pmc passes of this same command), else None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), key=lambda f: os.path.basename(f)[:3])
    for f in reversed(files):                       # newest round first (r03.json, r03_e2e.json: one per profiled configuration)
        for k, v in json.load(open(f)).items():
            if kernel + "(" in k and "hbm_read_bytes_corrected" in v:
                return v["hbm_read_bytes_corrected"] + v.get("hbm_write_bytes", 0.0), os.path.basename(f)
    return None, None


def cpu_baseline(C, L, max_threads=None):
    """One oracle training step at batch 16 on the host (bounded sample: ~15 s of CPU work on 16 threads)."""
    import torch
    from oracle import unet_ref
    from phasegen import detgen
    threads = max_threads or host_threads()
    torch.set_num_threads(threads)
    B = 16
    shapes = detgen.conv_shapes(C)
    g = torch.Generator().manual_seed(0)
    p = {}
    for k in detgen.param_order():
        if k in shapes:
            b = 1.0 / (shapes[k][1] * shapes[k][2]) ** 0.5
            p[k] = (torch.rand(shapes[k], generator=g) * 2 - 1) * b
        else:
            p[k] = torch.ones(2 * C) if k.endswith("weight") else torch.zeros(2 * C)
    stats = {}
    for k in detgen.BN_KEYS:
        stats[k + ".running_mean"] = torch.zeros(2 * C)
        stats[k + ".running_var"] = torch.ones(2 * C)
        stats[k + ".num_batches_tracked"] = torch.tensor(0)
    pp = dict(p)
    pp.update(stats)
    ost = unet_ref.new_opt_state(p)
    batch = torch.from_numpy(detgen.make_batch(B, C, L, seed=1))
    t0 = time.perf_counter()
    unet_ref.train_step(pp, batch, ost, stats)
    dt = time.perf_counter() - t0
    return {"value": B * L / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 full training step (fwd+loss+bwd+Adam) of the same C={C}, L={L} model at batch {B} "
                      f"({B * L} frames) by oracle/unet_ref.py (stock fp32 torch CPU ops on ATen's native conv path: oneDNN is OFF "
                      f"because its multi-threaded ConvTranspose1d is wrong at this size, DESIGN.md §2; measured in the build container, "
                      f"8 cores, forward at batch 2 x 128 frames: oneDNN on 2.12 s and wrong by 0.37, off 1.42 s -- the native path is "
                      f"not the slower one here), {dt:.1f} s wall"}


def synthetic_batch(torch, B, C, L, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    re = torch.randn(B, C, L, device="cuda", generator=gen)
    im = torch.randn(B, C, L, device="cuda", generator=gen)
    return torch.stack([torch.log1p(torch.sqrt(re * re + im * im)),
                        (torch.rand(B, C, L, device="cuda", generator=gen) * 2 - 1) * torch.pi], dim=1).contiguous()


_REAL_STDOUT = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to stdout when a communicator is created and gloo its
    rank chatter: route file descriptor 1 to stderr for the whole run and hand the line to the saved descriptor (emit)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(out):
    line = (json.dumps(out) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, line)
    else:
        os.write(_REAL_STDOUT, line)


def kernel_pass(torch, ops, step_fn, steps, fl, peak, step_ms):
    """Second pass: per-launch HIP events, grouped by the kernel symbol each label launches."""
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    ops.set_timer(None)
    ks, by = {}, {}
    hbm = {}
    for label, (n, ms) in sorted(timer.summary().items()):
        if label.startswith("hbm:"):              # BatchNorm / loss / Adam: algorithmic bytes over the launch's event time
            nb = timer.bytes.get(label, 0)
            grp = label[4:].split(".")[0]
            e = hbm.setdefault(grp, {"launches_per_step": 0, "ms_per_step": 0.0, "bytes_per_step": 0, "launches": {}})
            e["launches_per_step"] += 1
            e["ms_per_step"] += ms
            e["bytes_per_step"] += nb
            e["launches"][label[4:]] = {"ms": round(ms, 4), "bytes": nb, "TBps": round(nb / ms / 1e9, 3) if ms > 0 else None}
            continue
        plan = timer.plans.get(label, "?")
        kern = plan.split("|")[0]
        f = fl[label.split(".")[0]]
        ks[label] = {"launches": n, "ms": round(ms, 4), "tflops": round(f / ms / 1e9, 2), "kernel": kern,
                     "plan": plan.split("|", 1)[1] if "|" in plan else ""}
        e = by.setdefault(kern, {"launches_per_step": 0, "ms_per_step": 0.0, "flops_per_step": 0.0, "layers": []})
        e["launches_per_step"] += 1
        e["ms_per_step"] += ms
        e["flops_per_step"] += f
        e["layers"].append(label)
        if "|tail=" in plan:        # column tail handed to a second launch (conv_igemm.hip launch()): inside this label's time and flops
            e["tail_kernel"] = plan.split("|tail=")[1].split(",grid=")[0]
    for e in by.values():
        e["tflops"] = round(e["flops_per_step"] / e["ms_per_step"] / 1e9, 2)
        e["frac"] = round(e["tflops"] / peak, 4)
        e["share_of_step"] = round(e["ms_per_step"] / step_ms, 4)
        e["ms_per_step"] = round(e["ms_per_step"], 4)
    for e in hbm.values():
        e["TBps"] = round(e["bytes_per_step"] / e["ms_per_step"] / 1e9, 3) if e["ms_per_step"] > 0 else None
        e["frac_of_hbm_peak"] = round(e["TBps"] / PEAK_HBM_TBPS, 4) if e["TBps"] else None
        e["share_of_step"] = round(e["ms_per_step"] / step_ms, 4)
        e["ms_per_step"] = round(e["ms_per_step"], 4)
    kernel_pass.hbm = hbm         # (picked up by the caller right after the call)
    return ks, by


def roofline_of(by, peak, step_tflops, precision, headline_shape, alg_bytes=None):
    dom = max(by, key=lambda k: by[k]["ms_per_step"])
    big = {k: v for k, v in by.items() if v["share_of_step"] >= 0.003}
    worst = min(big, key=lambda k: big[k]["frac"])
    d = by[dom]
    traffic, src = pmc_traffic(dom) if headline_shape and precision == "fp32" else (None, None)
    if headline_shape:        # HBM-side bytes per launch (L2 fills + write-backs) of EVERY kernel, from the committed counter passes,
        for k in by:          # and their ratio to the algorithmic bytes of the launches the symbol covers (each operand once)
            t, tsrc = pmc_traffic(k)
            if t and by[k].get("tail_kernel"):       # the tail launch re-reads the weights: its counters belong to the same launches
                tt, _ = pmc_traffic(by[k]["tail_kernel"])
                t = t + tt if tt else None
            by[k]["traffic"], by[k]["traffic_source"] = t, tsrc
            if alg_bytes is not None:
                ab = sum(alg_bytes.get(l, 0) for l in by[k]["layers"]) / max(1, by[k]["launches_per_step"])
                by[k]["algorithmic_bytes_per_launch"] = ab
                by[k]["traffic_ratio"] = round(t / ab, 2) if (t and ab) else None
    return {"bound": "mfma", "kernel": dom + " (" + ", ".join(d["layers"]) + ")",
            "achieved": d["tflops"], "peak": peak, "unit": "TFLOP/s", "frac": d["tflops"] / peak,
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC: FETCH_SIZE x 2 + WRITE_SIZE)", "traffic_source": src,
            "launches_per_step": d["launches_per_step"], "ms_per_step_in_kernel": d["ms_per_step"], "share_of_step": d["share_of_step"],
            "flops_per_launch": d["flops_per_step"] / d["launches_per_step"],
            "ms_per_launch": d["ms_per_step"] / d["launches_per_step"],
            "step_tflops": round(step_tflops, 2), "step_frac": round(step_tflops / peak, 4),
            **({"sustained_peak": {"tflops": SUSTAINED_BF16_MFMA_TFLOPS_RANDOM, "what": "register-only v_mfma_f32_32x32x16_bf16 loop on "
                                   "uniform random operands, all SIMDs, no memory traffic: 1.78 GHz under the power limit "
                                   "(constant operands: 2460 TFLOP/s at 2.39 GHz)", "source": "profiles/r03_mfma_peak.txt",
                                   "frac": round(d["tflops"] / SUSTAINED_BF16_MFMA_TFLOPS_RANDOM, 4),
                                   "step_frac": round(step_tflops / SUSTAINED_BF16_MFMA_TFLOPS_RANDOM, 4)}}
               if peak == PEAK_BF16_MFMA_TFLOPS else {}),
            "worst_kernel": {"kernel": worst + " (" + ", ".join(by[worst]["layers"]) + ")", "tflops": by[worst]["tflops"],
                             "frac": by[worst]["frac"], "ms_per_step": by[worst]["ms_per_step"]},
            "by_kernel": by}


def dp_diagnostics(torch, dist, trainer, batch, world, step_ms):
    """N > 1 only.  When the replicas' parameters differ after the timed steps the line carries dp.error / "invalid" and the
    process exits with status 3 after printing it."""
    from phasegen.unet import BACKWARD_ORDER
    eng = trainer.engine
    flat = eng.arena.flat
    # (1) ranks that really take part in a collective, and identical parameters everywhere
    one = torch.ones(1, device="cuda")
    dist.all_reduce(one)
    chk = torch.zeros(2, device="cuda", dtype=torch.float64)
    for s0 in range(0, flat.numel(), 1 << 26):
        c = flat[s0:s0 + (1 << 26)].double()
        chk[0] += c.sum()
        chk[1] += (c * c).sum()
    got = [torch.empty_like(chk) for _ in range(world)]
    dist.all_gather(got, chk)
    same = all(torch.equal(got[0], t) for t in got)
    out = {"rccl_ranks": int(one.item()), "backend": dist.get_backend(), "replicas_identical": bool(same),
           "param_checksum": [float(v) for v in got[0].cpu()]}
    if not same:        # the run is INVALID: say so in the line (rank 0 still prints it) and exit non-zero after it
        out["error"] = "data-parallel replicas diverged"
        out["checksums"] = [[float(v) for v in t.cpu()] for t in got]
        return out
    # (2) every bucket's all-reduce alone on an otherwise idle chip (events on the launch stream, which waits for RCCL's)
    red = trainer.reducer
    alone = {}
    for rep in range(2):
        for name in BACKWARD_ORDER:
            g = red.buckets.view(name)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=red.group)
            b.record()
            torch.cuda.synchronize()
            alone[name] = a.elapsed_time(b)
    total = sum(alone.values())
    payload = sum(red.buckets.view(n).numel() for n in BACKWARD_ORDER) * 4
    # (3) the same step without communication
    red.enabled = False
    for _ in range(1):
        trainer.step(batch)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        trainer.step(batch)
    torch.cuda.synchronize()
    t = torch.tensor([(time.perf_counter() - t0) / 3 * 1e3], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    red.enabled = True
    nocomm = float(t.item())
    exposed = max(0.0, step_ms - nocomm)
    B, L = batch.shape[0], batch.shape[3]
    out["dp_equivalent"] = {"ms_per_step": round(nocomm, 3), "frames_per_s_per_gpu": B * L / nocomm * 1e3,
                            "efficiency": round(nocomm / step_ms, 4),
                            "note": "this run's own step with the all-reduce switched off (side-stream Adam, contended work split): "
                                    "efficiency = value / (n_gpus x frames_per_s_per_gpu); the N = 1 line carries the same step as "
                                    "other_configs.dp_equivalent"}
    out.update({"allreduce_alone_ms": {k: round(v, 3) for k, v in alone.items()}, "allreduce_alone_total_ms": round(total, 3),
                "allreduce_payload_bytes": payload,
                "allreduce_busbw_GBps": round(payload * 2 * (world - 1) / max(world, 1) / (total * 1e-3) / 1e9, 1),
                "step_ms_compute_only": round(nocomm, 3), "comm_exposed_ms": round(exposed, 3),
                "comm_hidden_frac": round(max(0.0, min(1.0, 1.0 - exposed / total)), 4) if total > 0 else None})
    return out


def run_train(a, torch, dist, world, rank, local):
    from phasegen import ops
    from phasegen.model import UNetModel
    from phasegen.trainer import Trainer
    C, L, B = a.channels, a.frames, a.batch
    peak = PEAK_FP32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    dev = device_info(torch) if rank == 0 else None
    torch.manual_seed(0)
    model = UNetModel(C, 2 * C, gpu_ids=[local], precision=a.precision)
    selftest = a.dp_selftest and world == 1         # one-rank RCCL group: every collective of the N > 1 path is really issued
    trainer = Trainer(model, lr=1e-3, grad_compress=None if a.grad_compress == "none" else a.grad_compress, always_reduce=selftest,
                      overlap_adam=not a.serial_adam, fuse_adam=not a.no_fused_adam)
    batch = synthetic_batch(torch, B, C, L, 1 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.step(batch)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = trainer.step(batch)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = [float(v) for v in losses.cpu()]
    step_ms = dt / a.steps * 1e3
    dp = dp_diagnostics(torch, dist, trainer, batch, world, step_ms) if (world > 1 or selftest) else None

    fl = conv_flops(C, L, B)
    step_flops = 3 * sum(fl.values()) - fl["D0"]
    ks, by = kernel_pass(torch, ops, lambda: trainer.step(batch), 3, fl, peak, step_ms)
    hbm, hbm_mode = kernel_pass.hbm, "the timed step's own launches"
    plain = None
    if trainer.fuse_adam:
        # the wgrad launches above carry the Adam update of their weight (24 B of HBM traffic per parameter in the epilogue, not
        # counted in `achieved`'s flops): time the same kernels once more WITHOUT it, so the line shows both
        trainer.fuse_adam = False
        trainer.step(batch)
        _, plain = kernel_pass(torch, ops, lambda: trainer.step(batch), 3, fl, peak, step_ms)
        # the HBM-bound side of the step (BatchNorm, loss, Adam) is reported from THIS pass: with the update fused into the wgrad
        # epilogues the headline step launches no Adam kernel for the conv weights at all; here it runs as per-layer slices on the side
        # stream, beside backward's convolutions -- what a data-parallel rank executes
        hbm, hbm_mode = kernel_pass.hbm, "pass with fuse_adam=False: Adam as per-layer slices on the side stream, beside backward's dgrad kernels"
        trainer.fuse_adam = True
    if rank != 0:
        return
    frames = world * B * L * a.steps
    out = {
        "metric": "spectrogram-frames/sec (train fwd+bwd)", "value": frames / dt, "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": step_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPES[a.precision], "data": "synthetic",
        "config": {"workload": f"train.py full step (fwd + cos/sin/mag loss + bwd + Adam{' + RCCL grad all-reduce' if world > 1 else ''}), "
                               f"UNetModel({C}, {2 * C}), per-GPU batch {B} x {C} bins x {L} frames (BASELINE configs[2]{'/[3]' if world > 1 else ''})",
                   "global_batch": world * B, "frames": L, "channels": C, "parallelism": f"dp{world}",
                   "grad_allreduce_payload": "fp32" if a.grad_compress == "none" else a.grad_compress,
                   "adam": ("fused into the wgrad epilogues" if trainer.fuse_adam else
                            ("per-layer slices on a side stream" if trainer.overlap_adam else "one launch after backward")),
                   "final_loss": loss_val},
        "roofline": roofline_of(by, peak, step_flops / (dt / a.steps) / 1e12, a.precision, (C, L, B) == (1024, 256, 64),
                                conv_bytes(C, L, B, trainer.fuse_adam)),
        "hbm_kernels": {"bound": "hbm", "peak_TBps": PEAK_HBM_TBPS, "from": hbm_mode,
                        "what": "algorithmic bytes (each operand once: BatchNorm forward 1 read + 1-2 writes, backward 2 reads + 1 write, "
                                "loss 16 B in + 8 B out per bin-frame, Adam 28 B per parameter) / HIP-event time of the launch on its stream",
                        "by_kernel": hbm},
        "kernels": ks,
        "device": dev,
    }
    if plain is not None:
        dom = out["roofline"]["kernel"].split(" (")[0]
        if dom in plain:
            nparam = {"D0": 2 * C * C * 32, "U0": 4 * C * 2 * C * 32}
            out["roofline"]["fused_update"] = {
                "what": "this kernel's launches also apply Adam to the weight they produce the gradient of (pg_conv_args.adam): "
                        "24 B of HBM traffic per parameter in the epilogue, inside the measured duration, outside `achieved`'s flops",
                "without_it": {"achieved": plain[dom]["tflops"], "frac": plain[dom]["frac"], "ms_per_step_in_kernel": plain[dom]["ms_per_step"]},
                "adam_bytes_per_step_in_kernel": sum(24 * nparam.get(l.split(".")[0], 0) for l in by[dom]["layers"]),
            }
    if dp is not None:
        out["dp"] = dp
    if dp is not None and not dp["replicas_identical"]:
        out["invalid"] = "data-parallel replicas diverged: value is not a valid measurement"
    if a.rehearse_on_one_gpu and world > 1:
        out["rehearsal"] = True
        out["rehearsal_note"] = f"{world} ranks share ONE GPU and all-reduce through the host (gloo): code-path check, not a measurement"
    if world == 1 and a.precision == "fp32" and not a.no_other_precisions:
        other = {}
        for mode in ("bf16x3", "bf16"):
            model.engine.precision = ops.precision_code(mode)
            for _ in range(2):
                trainer.step(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                trainer.step(batch)
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t1) / 5
            other[mode] = {"ms_per_step": d1 * 1e3, "frames_per_s": B * L / d1, "step_tflops": round(step_flops / d1 / 1e12, 2)}
        model.engine.precision = ops.precision_code("fp32")
        out["other_precisions"] = other
    if world == 1 and a.precision == "fp32" and not a.no_other_configs and (C, L, B) == (1024, 256, 64):
        out["other_configs"] = other_configs(torch, dist, model, C, L, B)      # (each leg catches its own failure)
        out["projected_dp_efficiency_8gpu"] = projected_dp_efficiency(out)
    if world == 1 and not a.no_cpu_baseline and a.precision == "fp32":
        del trainer, model, batch
        torch.cuda.empty_cache()
        try:
            out["cpu_baseline"] = cpu_baseline(C, L, a.cpu_threads)
        except Exception as e:          # noqa: BLE001
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    emit(out)
    if "invalid" in out:
        sys.stdout.flush()
        raise SystemExit(3)


def _timed(torch, dist, world, fn, warmup, steps):
    """`warmup` untimed calls, then `steps` calls bracketed by barrier + synchronize; max over ranks.  Seconds per call."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt / steps


def measure_fwd(torch, dist, world, rank, model, C, L, B, warmup, steps, precision="fp32"):
    """BASELINE configs[1]: forward only (train-mode BatchNorm, as the reference always runs it), batch 32 x 1024 x 256."""
    from phasegen import ops
    peak = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    x = synthetic_batch(torch, B, C, L, 1 + rank)[:, 0].contiguous()
    fwd = lambda: model.engine.forward(x, update_stats=True)
    sec = _timed(torch, dist, world, fwd, warmup, steps)
    fl = conv_flops(C, L, B)
    step_ms = sec * 1e3
    ks, by = kernel_pass(torch, ops, fwd, 3, fl, peak, step_ms)
    enc = sum(fl[n] for n in ("D0", "D1", "D2", "D3")) / sum(ks[n + ".fwd"]["ms"] for n in ("D0", "D1", "D2", "D3")) / 1e9
    return {
        "metric": "spectrogram-frames/sec (forward only)", "value": world * B * L / sec, "unit": "frames/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": step_ms, "higher_is_better": True,
        "scaling": "weak (replicas only)", "vs_baseline": None, "dtype": DTYPES[precision], "data": "synthetic",
        "config": {"workload": f"UNetModel({C}, {2 * C}).forward, train-mode BatchNorm, batch {B} x {C} bins x {L} frames (BASELINE configs[1])",
                   "global_batch": world * B, "frames": L, "channels": C},
        "roofline": roofline_of(by, peak, sum(fl.values()) / sec / 1e12, precision, False),
        "encoder_convs": {"tflops": round(enc, 2), "frac": round(enc / peak, 4),
                          "note": "D0-D3 forward together: BASELINE.json's '>= 50 % of the MFMA-fp32 roofline on the encoder convs'"},
        "kernels": ks}


def measure_e2e(torch, dist, world, rank, model, clips, warmup, steps, prec="bf16"):
    """BASELINE configs[4]: stereo clips (2 mono signals each) of 130 560 samples -> STFT 2048/512 fused with log1p|z| / angle ->
    U-Net forward (bf16-resident operands by default) -> ISTFT of (exp(m) - 1) e^{j phi}.  No exchange step: ranks are replicas."""
    from phasegen import audio, ops
    n_fft, hop, n = 2048, 512, 255 * 512
    dev = device_info(torch) if rank == 0 else None
    peak = PEAK_FP32_MFMA_TFLOPS if prec == "fp32" else PEAK_BF16_MFMA_TFLOPS
    C = n_fft // 2
    nsig, frames = 2 * clips, 1 + n // hop
    g = torch.Generator(device="cuda").manual_seed(1 + rank)
    wav = torch.randn(nsig, n, device="cuda", generator=g) * 0.1
    polar = torch.empty(nsig, 2, C, frames, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    stage = [0.0, 0.0, 0.0]
    res = {}

    def step(record=False):
        if record:
            ev[0].record()
        ops.stft(wav, n_fft, hop, polar=True, out=polar)
        if record:
            ev[1].record()
        pred = model.engine.forward(polar[:, 0], update_stats=False, inference=True)
        if record:
            ev[2].record()
        res["out"] = [audio.synthesize(polar[i:i + 64, 0], pred[i:i + 64, :C], hop) for i in range(0, nsig, 64)]
        if record:
            ev[3].record()
            torch.cuda.synchronize()
            for i in range(3):
                stage[i] += ev[i].elapsed_time(ev[i + 1])

    sec = _timed(torch, dist, world, step, warmup, steps)
    out = res["out"]
    assert out[0].shape == (min(64, nsig), hop * (frames - 1)) and bool(torch.isfinite(out[0]).all())
    for _ in range(3):
        step(True)
    fl = conv_flops(C, frames, nsig)
    step_ms = sec * 1e3
    # the forward stage once more, CLEAN (no events, no synchronisation between stages: the event-bracketed pass above exposes the
    # host's launch latency after every synchronize and reads ~10 % long); this is the time the roofline fraction is taken on
    fwd = lambda: model.engine.forward(polar[:, 0], update_stats=False, inference=True)
    fwd_ms = _timed(torch, dist, world, fwd, 2, max(steps, 5)) * 1e3
    graph_ms = None
    if prec == "bf16":      # the same forward replayed from a captured HIP graph: 8 convs + fixups + 6 BatchNorms, one launch
        model.engine.graphs = True
        try:
            graph_ms = _timed(torch, dist, world, fwd, 3, max(steps, 5)) * 1e3
        finally:
            model.engine.graphs = False
    ks, by = kernel_pass(torch, ops, fwd, 3, fl, peak, fwd_ms)
    roof = roofline_of(by, peak, sum(fl.values()) / (fwd_ms * 1e-3) / 1e12, prec, clips == 32)
    roof["step_frac_of"] = ("U-Net forward stage (input cast + 8 convs + 6 BatchNorms), timed clean: all conv FLOPs / stage time / peak; "
                            "by_kernel holds per-launch event times (an event pair per launch reads a few % longer than the clean loop)")
    dom = roof["kernel"].split(" (")[0]
    roof["traffic"], roof["traffic_source"] = pmc_traffic(dom)
    # HBM-side stages.  Algorithmic bytes per frame: STFT 2048/512 reads hop samples
    # (2 KB) and writes 1024 bins x [re; im] or [log1p|z|; angle] (8 KB); the ISTFT reads 8 KB and writes 2 KB (+ 2 x 2 KB for the
    # peak normalisation's read-modify-write, which the algorithm needs: the peak is known only after the last sample).
    def stage_roof(ms, nbytes, kernels):
        t = [pmc_traffic(k) for k in kernels]
        tr = sum(v for v, _ in t if v) if all(v for v, _ in t) else None
        return {"ms": round(ms, 4), "algorithmic_bytes": nbytes, "TBps": round(nbytes / ms / 1e9, 3), "frac": round(nbytes / ms / 1e9 / PEAK_HBM_TBPS, 4),
                "bound": "hbm", "peak_TBps": PEAK_HBM_TBPS, "traffic": tr, "traffic_ratio": round(tr / nbytes, 2) if tr else None,
                "traffic_source": t[0][1], "kernels": kernels}
    nf = nsig * frames
    stage_roofline = {"stft+polar": stage_roof(stage[0] / 3, nf * (hop * 4 + n_fft * 4), ["stft_w_kernel<false, 16>"]),
                      "istft": stage_roof(stage[2] / 3, nf * (n_fft * 4 + hop * 4 + 2 * hop * 4),
                                          ["istft_ola_w_kernel<16>", "istft_seam_kernel", "istft_peak_normalize_kernel"])}
    return {
        "metric": "spectrogram-frames/sec end to end (STFT + U-Net forward + ISTFT)", "value": world * nsig * frames / sec,
        "unit": "frames/s", "clips_per_s": world * clips / sec, "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak (replicas only)", "vs_baseline": None,
        "dtype": DTYPES[prec], "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4]: {clips} stereo clips x {n} samples per rank, 2048-FFT / 512-hop, "
                               f"STFT+polar -> UNetModel({C}, {2 * C}).forward -> ISTFT", "signals_per_rank": nsig, "frames": frames},
        "stage_ms": {"stft+polar": stage[0] / 3, "unet_forward": fwd_ms, "istft": stage[2] / 3,
                     "unet_forward_event_bracketed": stage[1] / 3, "unet_forward_graph_replay": graph_ms},
        "stage_roofline": stage_roofline, "roofline": roof, "kernels": ks, "device": dev}


def measure_train(torch, model, B, C, L, seed, fuse_adam, warmup, steps, contended=False, held=0, hold_shape="rccl+mem"):
    """A full train.py:41-62 step at another shape / update placement on the SAME model (N = 1): ms and frames/s only.
    ``contended``: backward's convolutions take the work split a data-parallel rank uses (engine.contended, set by the Trainer when
    world > 1).  ``held``: the steps run while a collective-shaped kernel (tools/spin/spin.hip: 256 threads, <= 113 VGPRs, 32 KB LDS
    per workgroup) holds that many CUs from a side stream, as RCCL's kernels do during a data-parallel backward.  ``hold_shape``:
    "rccl+mem" = its waves stream memory the whole time (what a collective's data movers do), "rccl" = they spin on dependent-free
    VALU work, taking every issue slot the conv wave on the same SIMD leaves (the worst case, not a collective's behaviour)."""
    from phasegen.trainer import Trainer
    trainer = Trainer(model, lr=1e-3, fuse_adam=fuse_adam)
    batch = synthetic_batch(torch, B, C, L, seed)
    eng = model.engine
    old = eng.contended
    eng.contended = bool(contended)
    info = None
    try:
        if not held:
            sec = _timed(torch, None, 1, lambda: trainer.step(batch), warmup, steps)
        else:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from hold import Hold
            for _ in range(warmup):
                trainer.step(batch)
            torch.cuda.synchronize()
            main = torch.cuda.current_stream()
            hold = Hold()
            hold.start(held, shape=hold_shape, max_us=int(1e6 * (2.0 + steps * (0.4 if hold_shape == "rccl+mem" else 2.0))))
            try:                        # nothing below may wait for the DEVICE (that would wait for the hold kernel): the main stream only
                t0 = time.perf_counter()
                for _ in range(steps):
                    trainer.step(batch)
                main.synchronize()      # (Trainer.step ends with the main stream waiting for its side stream)
                sec = (time.perf_counter() - t0) / steps
            finally:
                info = hold.stop()
            if info["held_ms_min"] < sec * steps * 1e3 * 0.98:
                info["warning"] = "the hold kernel left before the timed steps ended"
    finally:
        eng.contended = old
    fl = conv_flops(C, L, B)
    tf = (3 * sum(fl.values()) - fl["D0"]) / sec / 1e12
    out = {"ms_per_step": sec * 1e3, "frames_per_s": B * L / sec, "steps": steps, "warmup": warmup, "batch": B, "frames": L,
           "channels": C, "step_tflops": round(tf, 2), "step_frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
           "adam": "fused into the wgrad epilogues" if trainer.fuse_adam else "per-layer slices on a side stream",
           "work_split": "contended (data-parallel policy)" if contended else "automatic"}
    if info is not None:
        out["hold"] = info
    return out


def other_configs(torch, dist, model, C, L, B):
    """The other BASELINE configurations on the headline run's model (N = 1, fp32), each a few seconds.  A leg that fails reports
    its error under its own key: the others, and the headline, stand."""
    from phasegen import ops
    out = {}

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:          # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    def fwd():
        f = measure_fwd(torch, dist, 1, 0, model, C, L, 32, 3, 10)
        r = {k: f[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "encoder_convs")}
        r["roofline"] = {k: v for k, v in f["roofline"].items() if k != "by_kernel"}
        return r

    def e2e():
        old = model.engine.precision
        model.engine.precision = ops.precision_code("bf16")
        try:
            e = measure_e2e(torch, dist, 1, 0, model, 32, 3, 10)
        finally:
            model.engine.precision = old
        return {k: e[k] for k in ("metric", "value", "unit", "clips_per_s", "ms_per_step", "dtype", "config", "stage_ms", "stage_roofline",
                                  "roofline", "device")}

    leg("dp_equivalent", lambda: dict(measure_train(torch, model, B, C, L, 1, False, 2, 5, contended=True),
                                      note="the headline step as every rank of an N > 1 run executes it: Adam NOT fused into the wgrad epilogues "
                                           "(per-layer slices on a side stream) and engine.contended = True (the data-parallel work split)"))
    leg("dp_equivalent_held32", lambda: dict(measure_train(torch, model, B, C, L, 1, False, 2, 5, contended=True, held=32),
                                             note="the same step while a collective-shaped kernel (256 threads, <= 113 VGPRs, 32 KB LDS, streaming "
                                                  "memory) holds 32 CUs for its whole duration (tools/contention.py, profiles/r04_contention.json): a "
                                                  "one-GPU stand-in for RCCL's share of the chip -- pessimistic in time (a 2.45 GB all-reduce occupies "
                                                  "a fraction of backward, not the whole step)"))
    leg("dp_equivalent_held32_alu_spin", lambda: dict(measure_train(torch, model, B, C, L, 1, False, 1, 2, contended=True, held=32, hold_shape="rccl"),
                                                      note="worst case, NOT a collective's behaviour: the held workgroups spin on dependent-free VALU "
                                                           "work and take every issue slot the conv waves on their SIMDs leave"))
    leg("ref_default", lambda: dict(measure_train(torch, model, 16, C, 128, 3, True, 2, 10),
                                    note="the reference's own defaults, train.py:14-15: batch 16, 1024 bins x 128 frames"))
    leg("fwd", fwd)
    leg("e2e", e2e)
    leg("demo_clip", lambda: measure_demo_clip(torch, model, C))
    return out


def measure_demo_clip(torch, model, C):
    """demo.py's timed region (demo.py:33-45: one clip of 1024 bins x 128 frames -> forward -> (exp(m) - 1) e^{j phi} -> ISTFT -> host),
    per MFMA operand mode.  At batch 1 a forward is one pass over the 2.45 GB of weights (N = 65 ... 14 columns per layer): the conv
    launches are priced against HBM (weight bytes / time / 8 TB/s) next to their MFMA fraction."""
    from phasegen import audio, ops
    L = 128
    d = synthetic_batch(torch, 1, C, L, 5)
    fl = conv_flops(C, L, 1)
    wbytes = {"fp32": 4.0, "bf16": 2.0}
    out = {"workload": f"demo.py:33-45: 1 clip x {C} bins x {L} frames, forward + ISTFT + device->host", "weights": sum(
        int(v.numel()) for k, v in ((k, model.engine.arena.p(k)) for k in model.engine.arena.shapes) if v.dim() == 3)}
    old = model.engine.precision
    try:
        for mode in ("fp32", "bf16"):
            model.engine.precision = ops.precision_code(mode)

            def clip():
                with torch.no_grad():
                    pred = model.forward(d[:, 0])
                    return audio.synthesize(d[:, 0], pred[:, :C], 512).cpu()
            for _ in range(3):
                clip()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                clip()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            timer = ops.KernelTimer()
            ops.set_timer(timer)
            for _ in range(3):
                clip()
            torch.cuda.synchronize()
            ops.set_timer(None)
            conv = {k: v[1] for k, v in timer.summary().items() if not k.startswith("hbm:")}
            conv_ms = sum(conv.values())
            nb = out["weights"] * wbytes[mode]
            peak = PEAK_FP32_MFMA_TFLOPS if mode == "fp32" else PEAK_BF16_MFMA_TFLOPS
            out[mode] = {"ms_per_clip": round(ms, 4), "conv_launches_ms": round(conv_ms, 4), "weight_bytes": nb,
                         "weight_TBps": round(nb / conv_ms / 1e9, 3), "frac_of_hbm_peak": round(nb / conv_ms / 1e9 / PEAK_HBM_TBPS, 4),
                         "conv_tflops": round(sum(fl.values()) / conv_ms / 1e9, 1), "frac_of_mfma_peak": round(sum(fl.values()) / conv_ms / 1e9 / peak, 4),
                         "by_layer_ms": {k: round(v, 4) for k, v in sorted(conv.items())},
                         "plans": {k: timer.plans.get(k, "?").split("|")[0] for k in sorted(conv)}}
    finally:
        model.engine.precision = old
    return out


def projected_dp_efficiency(out):
    """dp_equivalent_held32 / headline: what one GPU can say about the N > 1 step (no multi-GPU node is reachable from the builder)."""
    oc = out.get("other_configs", {})
    h = oc.get("dp_equivalent_held32", {})
    if "frames_per_s" in h:
        return {"value": round(h["frames_per_s"] / out["value"], 4),
                "what": "frames/s of the data-parallel step (side-stream Adam, contended split) beside a collective-shaped kernel on 32 CUs "
                        "/ frames/s of the headline step; excludes the wire time of the 2.45 GB all-reduce itself (DESIGN.md section 4.5)"}
    return None


def run_fwd(a, torch, dist, world, rank, local):
    from phasegen.model import UNetModel
    C, L, B = a.channels, a.frames, (a.batch if a.batch != 64 else 32)
    torch.manual_seed(0)
    model = UNetModel(C, 2 * C, gpu_ids=[local], precision=a.precision)
    out = measure_fwd(torch, dist, world, rank, model, C, L, B, a.warmup, a.steps, a.precision)
    if rank == 0:
        emit(out)


def run_e2e(a, torch, dist, world, rank, local):
    from phasegen.model import UNetModel
    prec = a.precision if a.precision_given else "bf16"
    torch.manual_seed(0)
    model = UNetModel(1024, 2048, gpu_ids=[local], precision=prec)
    out = measure_e2e(torch, dist, world, rank, model, (a.batch if a.batch != 64 else 32), a.warmup, a.steps, prec)
    if rank == 0:
        emit(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["train", "fwd", "e2e"], default="train",
                    help="train = the headline (BASELINE configs[2]/[3]); fwd = configs[1]; e2e = configs[4]")
    ap.add_argument("--channels", type=int, default=1024, help="C (bins); 1024 = the reference's hard-coded model")
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (fwd / e2e default: 32 samples / 32 stereo clips)")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3"], default=None,
                    help="MFMA operand precision: fp32 = the parity path and the headline (default; e2e defaults to bf16 = BASELINE "
                         "configs[4]'s arithmetic: bf16 operands, fp32 accumulate, fp32 tensors and master weights)")
    ap.add_argument("--grad-compress", choices=["none", "bf16"], default="none",
                    help="N > 1: payload of the gradient all-reduce (bf16 halves the xGMI bytes; default fp32)")
    ap.add_argument("--dp-selftest", action="store_true",
                    help="N = 1 only: run the data-parallel code path (bucketed RCCL all-reduce, diagnostics) on a one-rank group")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 on a ONE-GPU box: every rank drives cuda:0 and the process group is gloo (device tensors staged through "
                         "the host).  Exercises the whole N > 1 code path (sharded seeds, bucketed async all-reduce from inside backward, "
                         "barrier + max-over-ranks timing, replica checksum, dp diagnostics); the JSON line carries \"rehearsal\": true "
                         "and its value is NOT a measurement")
    ap.add_argument("--no-fused-adam", action="store_true",
                    help="A/B: N = 1 runs the Adam update of every conv weight inside its wgrad kernel's epilogue; this flag falls back to "
                         "per-layer slices of a separate Adam kernel on a side stream (what N > 1 ranks always run)")
    ap.add_argument("--serial-adam", action="store_true", help="A/B: one Adam launch after backward instead of per-layer slices on a side stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the fwd / e2e / ref_default / dp_equivalent legs of the default line")
    ap.add_argument("--cpu-threads", type=int, default=None)
    a = ap.parse_args()
    a.precision_given = a.precision is not None
    if a.precision is None:
        a.precision = "fp32"

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or a.dp_selftest:
        quiet_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        # convenience: re-launch under torch.distributed.run as a CHILD process (never exec after GPU init)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), __file__] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the phasegen hot path has no CPU fallback")
    if a.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1 and a.rehearse_on_one_gpu:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif a.dp_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    {"train": run_train, "fwd": run_fwd, "e2e": run_e2e}[a.config](a, torch, dist, world, rank, local)
    if world > 1 or a.dp_selftest:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

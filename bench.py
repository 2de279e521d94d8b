#!/usr/bin/env python3
"""bench.py -- spectrogram-frames/sec of the full training step (train.py:41-62: fwd + loss + bwd + Adam
[+ RCCL gradient all-reduce for N > 1]) of the C=1024 U-Net on synthetic (64, 2, 1024, 256) batches per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  Extra objects:
  roofline     dominant kernel (U0 forward = conv_raw_kernel<32,2,true,0,2>, the largest single GEMM: 4.43 TFLOP per launch at
               batch 64) -- algorithmic FLOPs per launch / its average launch duration measured here with HIP events
               on the launch stream, against the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline the oracle (CPU restatement of the reference, oracle/unet_ref.py) doing the SAME training step on the
               host cores on a bounded sample (batch 16 of the same C=1024, L=256 model); rank 0, N = 1 only.
  kernels      per-layer conv timings (ms, TFLOP/s) for DESIGN.md's table.
  other_precisions  (N = 1, default fp32 run only) the same step re-timed for 5 steps in the two optional MFMA operand modes
               (pg_conv_set_precision): "bf16x3" = fp32 operands split into hi + lo bf16, 3 bf16 MFMA products (~5e-6 from
               exact, passes the golden parity suite) and "bf16" = operands rounded to bf16 (BASELINE configs[4]).  Reported
               beside the headline, never as `value`.
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

DOMINANT_KERNEL = "conv_raw_kernel<32, 2, true, 0, 2>"   # the symbol rocprofv3 reports for the fp32 U0 forward launch
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense bf16: 16 x the fp32 rate (v_mfma_f32_32x32x16_bf16: 32 cycles for 32 768 FLOP)


def conv_flops(C, L, B):
    """Algorithmic FLOPs (2 x MAC, padding taps counted -- SURVEY.md §8d) per launch of every conv pass."""
    from phasegen.unet import frame_plan
    L1, L2, L3, L4 = frame_plan(L)
    # name: (Cin, Cout, k, positions the taps are applied at)  conv: output frames; convT: input frames
    g = {"D0": (C, 2 * C, 32, L1), "D1": (2 * C, 2 * C, 8, L2), "D2": (2 * C, 2 * C, 8, L3), "D3": (2 * C, 4 * C, 4, L4),
         "U3": (4 * C, 2 * C, 5, L4), "U2": (4 * C, 2 * C, 8, L3), "U1": (4 * C, 2 * C, 8, L2), "U0": (4 * C, 2 * C, 32, L1)}
    return {n: 2.0 * B * pos * ci * co * k for n, (ci, co, k, pos) in g.items()}


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


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/rNN_pmc_summary.json:
    FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate --pmc passes of this same command), else None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    for k, v in d.items():
        if kernel_substr in k and "hbm_read_bytes_corrected" in v:
            return v["hbm_read_bytes_corrected"] + v.get("hbm_write_bytes", 0.0)
    return None


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
                      f"({B * L} frames) by oracle/unet_ref.py (stock fp32 torch CPU ops, oneDNN off: see DESIGN.md), "
                      f"{dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=1024, help="C (bins); 1024 = the reference's hard-coded model")
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--precision", choices=["fp32", "bf16", "bf16x3"], default="fp32",
                    help="MFMA operand precision: fp32 = the parity path and the headline; bf16 = BASELINE configs[4]'s arithmetic "
                         "(bf16 operands, fp32 accumulate, fp32 tensors and master weights) -- reported separately")
    ap.add_argument("--grad-compress", choices=["none", "bf16"], default="none",
                    help="N > 1: payload of the gradient all-reduce (bf16 halves the xGMI bytes; default fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precisions", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=None)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from phasegen import ops
    from phasegen.model import UNetModel
    from phasegen.trainer import Trainer

    C, L, B = a.channels, a.frames, a.batch
    ops.set_conv_precision(a.precision)
    peak = PEAK_FP32_MFMA_TFLOPS if a.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    torch.manual_seed(0)
    model = UNetModel(C, 2 * C, gpu_ids=[local])
    trainer = Trainer(model, lr=1e-3, grad_compress=None if a.grad_compress == "none" else a.grad_compress)
    gen = torch.Generator(device="cuda").manual_seed(1 + rank)
    re = torch.randn(B, C, L, device="cuda", generator=gen)
    im = torch.randn(B, C, L, device="cuda", generator=gen)
    batch = torch.stack([torch.log1p(torch.sqrt(re * re + im * im)),
                         (torch.rand(B, C, L, device="cuda", generator=gen) * 2 - 1) * torch.pi], dim=1).contiguous()
    del re, im

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        trainer.step(batch)
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = trainer.step(batch)
    sync()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = [float(v) for v in losses.cpu()]

    if rank == 0:
        fl = conv_flops(C, L, B)
        ks = {}
        for label, (n, ms) in sorted(timer.summary().items()):
            ks[label] = {"launches": n, "ms": round(ms, 4), "tflops": round(fl[label.split(".")[0]] / ms / 1e9, 2)}
        dom = ks["U0.fwd"]
        frames = world * B * L * a.steps
        out = {
            "metric": "spectrogram-frames/sec (train fwd+bwd)", "value": frames / dt, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 operands / f32 accumulate",
                      "bf16x3": "f32 operands split hi+lo into 3 bf16 MFMA products / f32 accumulate"}[a.precision], "data": "synthetic",
            "config": {"workload": f"train.py full step (fwd + cos/sin/mag loss + bwd + Adam{' + RCCL grad all-reduce' if world > 1 else ''}), "
                                   f"UNetModel({C}, {2 * C}), per-GPU batch {B} x {C} bins x {L} frames (BASELINE configs[2]{'/[3]' if world > 1 else ''})",
                       "global_batch": world * B, "frames": L, "channels": C, "parallelism": f"dp{world}",
                       "grad_allreduce_payload": "fp32" if a.grad_compress == "none" else a.grad_compress,
                       "final_loss": loss_val},
            "roofline": {"bound": "mfma", "kernel": DOMINANT_KERNEL + " (U0 forward, ConvTranspose1d 4096->2048 k32 s2)",
                         "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s",
                         "frac": dom["tflops"] / peak,
                         "traffic": pmc_traffic(DOMINANT_KERNEL) if (C, L, B, a.precision) == (1024, 256, 64, "fp32") else None,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/)",
                         "flops_per_launch": fl["U0"], "ms_per_launch": dom["ms"],
                         "step_tflops": round((3 * sum(fl.values()) - fl["D0"]) / (dt / a.steps) / 1e12, 2)},
            "kernels": ks,
        }
        if world == 1 and a.precision == "fp32" and not a.no_other_precisions:
            other = {}
            for mode in ("bf16x3", "bf16"):
                ops.set_conv_precision(mode)
                for _ in range(2):
                    trainer.step(batch)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    trainer.step(batch)
                torch.cuda.synchronize()
                d1 = (time.perf_counter() - t1) / 5
                other[mode] = {"ms_per_step": d1 * 1e3, "frames_per_s": B * L / d1,
                               "step_tflops": round((3 * sum(fl.values()) - fl["D0"]) / d1 / 1e12, 2)}
            ops.set_conv_precision("fp32")
            out["other_precisions"] = other
        if world == 1 and not a.no_cpu_baseline and a.precision == "fp32":
            del trainer, model, batch
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(C, L, a.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

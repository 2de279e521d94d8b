#!/usr/bin/env python3
"""BASELINE configs[4]: stereo MUSDB-shape clips end to end on the device --

    waveform (2 channels x 130 560 samples = 255 hops of 512) -> STFT 2048/512 fused with log1p|z| / angle
    -> U-Net forward (bf16 MFMA operands, fp32 accumulate, fp32 weights; --precision fp32 for the parity arithmetic)
    -> ISTFT of (exp(m) - 1) e^{j phi_pred} (demo.py:39-40) -> waveform.

Stereo = two independent mono items (preproc_mdb.py:112 loads mono), so B clips are 2B signals.
The path has no exchange step: under torch.distributed.run every rank processes its own clips ("replicas only").

    python tools/e2e_bench.py [--clips 32] [--precision bf16] [--steps 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/e2e_bench.py ...
Prints one JSON line (rank 0): frames/s and clips/s over all ranks, per-stage milliseconds (HIP events).
"""
import argparse, json, os, sys, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd")); sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=32, help="stereo clips per rank and step")
    ap.add_argument("--precision", choices=["bf16", "bf16x3", "fp32"], default="bf16")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from phasegen import ops, audio
    from phasegen.model import UNetModel
    n_fft, hop, n = 2048, 512, 255 * 512
    C, nsig = n_fft // 2, 2 * a.clips
    frames = 1 + n // hop
    ops.set_conv_precision(a.precision)
    torch.manual_seed(0)
    model = UNetModel(C, 2 * C, gpu_ids=[local])
    g = torch.Generator(device="cuda").manual_seed(1 + rank)
    wav = torch.randn(nsig, n, device="cuda", generator=g) * 0.1
    polar = torch.empty(nsig, 2, C, frames, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    stage = [0.0, 0.0, 0.0]

    def step(record):
        if record: ev[0].record()
        ops.stft(wav, n_fft, hop, polar=True, out=polar)
        if record: ev[1].record()
        pred = model.engine.forward(polar[:, 0], update_stats=False)
        if record: ev[2].record()
        outs = [audio.synthesize(polar[i:i + 64, 0], pred[i:i + 64, :C], hop) for i in range(0, nsig, 64)]
        if record:
            ev[3].record(); torch.cuda.synchronize()
            for i in range(3): stage[i] += ev[i].elapsed_time(ev[i + 1])
        return outs

    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step(True)
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t)
    assert out[0].shape == (min(64, nsig), hop * (frames - 1)) and bool(torch.isfinite(out[0]).all())
    if rank == 0:
        print(json.dumps({"metric": "spectrogram-frames/sec end to end (STFT + U-Net forward + ISTFT)",
                          "value": world * nsig * frames * a.steps / dt, "unit": "frames/s", "clips_per_s": world * a.clips * a.steps / dt,
                          "n_gpus": world, "steps": a.steps, "ms_per_step": dt / a.steps * 1e3, "scaling": "weak (replicas only)",
                          "dtype": {"bf16": "bf16 operands / f32 accumulate", "bf16x3": "f32 split into 3 bf16 MFMA products / f32 accumulate", "fp32": "f32"}[a.precision], "data": "synthetic",
                          "config": {"workload": f"BASELINE configs[4]: {a.clips} stereo clips x 130560 samples per rank, 2048-FFT / 512-hop",
                                     "signals_per_rank": nsig, "frames": frames},
                          "stage_ms": {"stft+polar": stage[0] / a.steps, "unet_forward": stage[1] / a.steps, "istft": stage[2] / a.steps}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Counterpart of the reference's ``train.py`` (a script with module-level constants, train.py:11-27) on MI355X.

Same defaults (log_dir "unet_llr/", batch 16, lr 1e-3, dataset/Pop_audio_{train,val}.npy, checkpoint every 4000 steps as
``ckpt_{cnt}``, the per-epoch line ``Epoch {} done, {} elasped, mag loss: {}, ang loss: {}``); every constant can be
overridden on the command line.  One step = train.py:41-62 through ``phasegen.trainer.Trainer`` (no host<->device
traffic inside the step; losses are accumulated on the device and read once per epoch, where the reference
synchronises every step at train.py:64-65).

Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N train.py ...``; one process per GPU,
RCCL all-reduce of gradients, each rank reads clips rank::world of every epoch's permutation.

Out of scope here: the TensorBoard validation dump of train.py:69-124 -- a JSON-lines log is
written instead.
"""
import argparse
import json
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser(description="Train the U-Net phase generator (reference train.py counterpart).")
    ap.add_argument("--log_dir", default="unet_llr/")                         # train.py:11
    ap.add_argument("--gpu", type=int, default=None, help="device index (reference: gpu_id = 2); default LOCAL_RANK or 0")
    ap.add_argument("--batch_size", type=int, default=16)                     # train.py:14
    ap.add_argument("--channels", type=int, default=1024)                     # train.py:15 UNetModel(1024, 2048)
    ap.add_argument("--train", default="dataset/Pop_audio_train.npy")         # train.py:19
    ap.add_argument("--lr", type=float, default=0.001)                        # train.py:26
    ap.add_argument("--max_steps", type=int, default=0, help="stop after this many steps (0 = run forever like the reference)")
    ap.add_argument("--ckpt_every", type=int, default=4000)                   # train.py:126
    ap.add_argument("--val", default="dataset/Pop_audio_val.npy")             # train.py:23
    ap.add_argument("--val_every", type=int, default=2000)                    # train.py:69
    ap.add_argument("--val_clips", type=int, default=3)                       # train.py:24 batch_size=3
    ap.add_argument("--gl_iters", type=int, default=250)                      # train.py:101
    ap.add_argument("--hop", type=int, default=512)
    ap.add_argument("--n_fft", type=int, default=2048)
    ap.add_argument("--resume", default=None, help="checkpoint written by a previous run (ckpt_N + ckpt_N.optim)")
    ap.add_argument("--synthetic", type=int, default=0, help="train on N synthetic clips instead of --train (no dataset ships)")
    ap.add_argument("--frames", type=int, default=128, help="frames per synthetic clip")
    ap.add_argument("--grad_compress", choices=["none", "bf16"], default="none", help="payload of the data-parallel gradient all-reduce")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "bf16"], default="fp32",
                    help="MFMA operand mode of the convolutions (pg_conv_args.precision): fp32 = the reference's arithmetic")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from phasegen import detgen
    from phasegen.data import SpectrogramLoader, get_fft_npy_loader
    from phasegen.model import UNetModel
    from phasegen.trainer import Trainer
    from phasegen.validate import validation_metrics

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    gpu_id = a.gpu if a.gpu is not None else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(gpu_id)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", gpu_id))
    torch.manual_seed(0)                                                      # same init and same permutations on every rank
    model = UNetModel(a.channels, a.channels * 2, gpu_ids=[gpu_id], precision=a.precision).cuda(gpu_id)
    if a.synthetic:
        d = torch.from_numpy(detgen.make_batch(a.synthetic, a.channels, a.frames, seed=1)).cuda()
        loader = SpectrogramLoader(d, torch.zeros(a.synthetic, 1, device=d.device), a.batch_size, True, rank, world, seed=0)
    else:
        loader = get_fft_npy_loader([a.train], [0, 1], batch_size=a.batch_size, precon=True, rank=rank, world=world, seed=0)
    trainer = Trainer(model, lr=a.lr, grad_compress=None if a.grad_compress == "none" else a.grad_compress)
    if a.resume:
        trainer.load_checkpoint(a.resume)
    val_loader = None
    if not a.synthetic and os.path.exists(a.val):           # every rank: validation is sharded over the ranks (validate.py)
        val_loader = get_fft_npy_loader([a.val], [0, 1], batch_size=a.val_clips, precon=True, seed=0)
    os.makedirs(a.log_dir, exist_ok=True)
    log = open(os.path.join(a.log_dir, "log.jsonl"), "a") if rank == 0 else None

    j = cnt = 0
    ang_sum = torch.zeros((), device="cuda")
    mag_sum = torch.zeros((), device="cuda")
    n_loss = 0
    while True:
        start = time.time()
        frames = 0
        for i, d in enumerate(loader):
            if d[0].size(0) < a.batch_size:                                   # train.py:38-39
                continue
            cnt += 1
            losses = trainer.step(d[0])
            ang_sum += losses[1]
            mag_sum += losses[2]
            n_loss += 1
            frames += d[0].size(0) * d[0].size(3) * world
            if cnt % a.val_every == 0 and val_loader is not None:              # train.py:69-124 (metrics only, JSONL)
                # at a step boundary, on EVERY rank (same seed -> same first batch everywhere): rank r takes clips r::W, the sums
                # are all-reduced -- nobody waits in the next gradient all-reduce while rank 0 runs 250 Griffin-Lim iterations
                vm = validation_metrics(model, val_loader.__iter__().__next__()[0], a.hop, a.n_fft, a.gl_iters, shard=world > 1)
                vm["steps"] = cnt
                if rank == 0:
                    log.write(json.dumps(vm) + "\n")
                    log.flush()
            if cnt % a.ckpt_every == 0 and rank == 0:
                trainer.save_checkpoint(a.log_dir + "/ckpt_{}".format(cnt))   # train.py:126-127 (+ optimiser state)
            if a.max_steps and cnt >= a.max_steps:
                break
        j += 1
        torch.cuda.synchronize()
        el = time.time() - start
        mag, ang = float(mag_sum) / max(n_loss, 1), float(ang_sum) / max(n_loss, 1)
        if rank == 0:
            print("Epoch {} done, {} elasped, mag loss: {}, ang loss: {}".format(j, el, mag, ang))   # train.py:130
            log.write(json.dumps({"epoch": j, "steps": cnt, "seconds": el, "frames_per_s": frames / max(el, 1e-9),
                                  "Ang Loss": ang, "Mag Loss": mag}) + "\n")
            log.flush()
        if a.max_steps and cnt >= a.max_steps:
            break
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

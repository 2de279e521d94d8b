"""GPU: the train.py / demo.py counterparts run end to end on synthetic data with the reference's file names and
printed lines (H2), and a checkpoint written by training loads into the demo."""
import os
import subprocess
import sys

import numpy as np
import pytest

from phasegen import detgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unet-phasegen_amd")


def run(args, cwd):
    r = subprocess.run([sys.executable] + args, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_train_then_demo(tmp_path):
    from oracle import signal_ref
    C, L, n_fft, hop = 16, 24, 32, 8
    os.makedirs(tmp_path / "dataset")
    clips = [detgen.make_clip(hop * (L - 1), seed=70 + i) for i in range(6)]
    arr = np.stack([signal_ref.chunk_and_stft(c, n_fft, hop) for c in clips]).astype(np.float32)     # (N, 2, 16, 24) as preproc writes
    np.save(tmp_path / "dataset" / "Pop_audio_train.npy", arr)
    np.save(tmp_path / "dataset" / "Pop_audio_val.npy", arr[:3])
    out = run([os.path.join(PKG, "train.py"), "--channels", str(C), "--batch_size", "2", "--max_steps", "6", "--ckpt_every", "4",
               "--log_dir", "unet_llr/", "--val_every", "3", "--gl_iters", "5", "--hop", str(hop), "--n_fft", str(n_fft)], cwd=str(tmp_path))
    assert "start loading" in out and "Epoch 1 done," in out and "mag loss:" in out and "ang loss:" in out
    ckpt = tmp_path / "unet_llr" / "ckpt_4"
    assert ckpt.exists() and (tmp_path / "unet_llr" / "ckpt_4.optim").exists()
    import json
    lines = [json.loads(l) for l in open(tmp_path / "unet_llr" / "log.jsonl")]
    val = [l for l in lines if "LMSE" in l]
    assert len(val) == 2 and all(np.isfinite([v["MSE"], v["NOPMSE"], v["LMSE"]]).all() for v in val)   # steps 3 and 6
    # resume from step 4 and continue
    out2 = run([os.path.join(PKG, "train.py"), "--channels", str(C), "--batch_size", "2", "--max_steps", "2", "--log_dir", "unet_llr2/",
                "--resume", str(ckpt), "--val_every", "1000"], cwd=str(tmp_path))
    assert "Epoch 1 done," in out2
    out = run([os.path.join(PKG, "demo.py"), "--genre", "Pop", "--n_songs", "2", "--n_fft", str(n_fft), "--hop", str(hop),
               "--weight", str(ckpt), "--channels", str(C)], cwd=str(tmp_path))
    assert "UNet - avg" in out and "sec per clip." in out and "GL - avg" in out and "sec per clip" in out
    from scipy.io import wavfile
    for c in range(2):
        sr, a = wavfile.read(tmp_path / "demo" / f"unet_Pop_{c}.wav")
        assert sr == 16000 and a.dtype == np.float32 and a.shape == (hop * (L - 1),) and abs(np.max(np.abs(a)) - 1) < 1e-5
        sr, gl = wavfile.read(tmp_path / "demo" / f"gl_Pop_{c}.wav")                  # demo.py:58 Griffin-Lim comparator
        assert gl.shape == a.shape and abs(np.max(np.abs(gl)) - 1) < 1e-5


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_train_accepts_the_precision_modes(tmp_path, precision):
    """--precision selects the MFMA operand mode (pg_conv_args.precision, carried by the engine) for the whole run; the losses of the first
    steps stay close to the fp32 run's (same synthetic clips, same initialisation)."""
    import re
    args = [os.path.join(PKG, "train.py"), "--channels", "16", "--batch_size", "2", "--max_steps", "3", "--synthetic", "4",
            "--frames", "24", "--val_every", "1000", "--ckpt_every", "1000", "--log_dir", "run/"]
    ref = run(args, cwd=str(tmp_path))
    got = run(args + ["--precision", precision], cwd=str(tmp_path))
    pat = re.compile(r"mag loss: ([0-9.eE+-]+)")
    a, b = [float(x) for x in pat.findall(ref)], [float(x) for x in pat.findall(got)]
    assert len(a) == len(b) > 0
    tol = 1e-4 if precision == "bf16x3" else 3e-2
    assert np.allclose(a, b, rtol=tol), (a, b)


def test_bench_contract_line_and_dp_selftest(tmp_path):
    """bench.py prints ONE JSON line with the contract's keys; `roofline` names the kernel with the largest time share and
    carries step_frac / worst_kernel / by_kernel; --dp-selftest drives the data-parallel code path (bucketed RCCL all-reduce
    from inside backward, parameter checksum across ranks, per-bucket timings) on a one-rank group."""
    import json
    out = run([os.path.join(ROOT, "bench.py"), "--channels", "32", "--frames", "64", "--batch", "4", "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline", "--no-other-precisions", "--dp-selftest"], cwd=str(tmp_path))
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.strip() == lines[0], out[:400]     # stdout holds the line and nothing else (RCCL's banner goes to stderr)
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "kernels"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f32" and d["unit"] == "frames/s" and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and 0 < r["step_frac"] < 1 and r["worst_kernel"]["frac"] <= r["frac"] + 1e-9
    dom = max(r["by_kernel"].values(), key=lambda v: v["ms_per_step"])
    assert r["kernel"].startswith(next(k for k, v in r["by_kernel"].items() if v is dom))         # the time-dominant kernel
    assert len(d["kernels"]) == 23 and all("kernel" in v and v["kernel"].startswith("conv_") for v in d["kernels"].values())
    dp = d["dp"]
    assert dp["rccl_ranks"] == 1 and dp["backend"] == "nccl" and dp["replicas_identical"] is True
    assert set(dp["allreduce_alone_ms"]) == {"U0", "U1", "U2", "U3", "D3", "D2", "D1", "D0"} and dp["step_ms_compute_only"] > 0


def test_bench_two_ranks_rehearsed_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), except that both
    ranks share cuda:0 and all-reduce over gloo (--rehearse-on-one-gpu; RCCL refuses two ranks on one device): the N > 1
    branch -- per-rank seeds, bucketed async all-reduce inside backward, barrier + MAX-over-ranks timing, the replica checksum
    that fails the run when ranks diverge, the dp diagnostics -- runs for real, and rank 0 alone prints the line."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--channels", "32", "--frames", "64",
               "--batch", "4", "--steps", "2", "--warmup", "1", "--rehearse-on-one-gpu"], cwd=str(tmp_path))
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.strip() == lines[0], out[:400]     # ... also with two ranks (gloo's rank chatter, the other rank's prints)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rehearsal"] is True and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 2 * 4 * 64 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]        # whole-job frames / max-over-ranks time
    dp = d["dp"]
    assert dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and dp["replicas_identical"] is True
    assert dp["allreduce_payload_bytes"] > 0 and 0.0 <= dp["comm_hidden_frac"] <= 1.0
    assert "cpu_baseline" not in d and "other_precisions" not in d                              # N = 1 only

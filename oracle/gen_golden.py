"""ORACLE tooling -- generate tests/golden/*.npz by importing the reference (build container only).

Run:  python oracle/gen_golden.py [--full]

What it does (SURVEY.md §8c fixture list):
  G1/G2/G3  imported reference ``model.UNetModel(C, 2C, norm_layer=nn.BatchNorm1d, gpu_ids=[])`` on CPU,
            deterministic weights/inputs from phasegen.detgen: forward (+ per-layer intermediates via
            hooks on the oracle side), gradients of the train.py:45-60 loss, three torch.optim.Adam steps.
  G4        imported reference ``data.get_spec_and_angle`` on a seeded array with edge cases.
  G5        demo triple (logmag -> pred -> audio) at C=16 (ISTFT part is the oracle's: parity unpinned).
  G6        (--full) one full-size C=1024, L=128, B=1 forward: per-layer statistics + 4096 sampled outputs.
While generating it asserts that oracle/unet_ref.py reproduces the reference, which is what pins the oracle.

The reference source is only imported, never copied; the fixtures hold inputs' seeds and outputs only.
``librosa`` (absent) is stubbed with empty modules so that ``import model`` succeeds (model.py:7 pulls
unused names from utils.py, which imports librosa at module scope).
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "unet-phasegen_amd"))
sys.path.insert(0, ROOT)

from phasegen import detgen  # noqa: E402
from oracle import unet_ref, signal_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    for m in ("librosa", "librosa.display"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, REF)
    import model as refmodel  # noqa
    import data as refdata  # noqa
    sys.path.remove(REF)
    return refmodel, refdata


def ref_model(refmodel, C, params_np):
    m = refmodel.UNetModel(C, 2 * C, norm_layer=nn.BatchNorm1d, gpu_ids=[])
    sd = {k: torch.from_numpy(np.array(v)) for k, v in params_np.items()}
    m.model.load_state_dict(sd)
    m.train()
    return m


def ref_loss(pred, batch, C):
    """train.py:45-60 composed from stock torch, exactly as written there."""
    lossf = torch.nn.MSELoss()
    pred_p, pred_m = pred[:, :C], pred[:, C:]
    cos_loss = lossf(torch.cos(pred_p), batch[:, 1, ...].cos())
    sin_loss = lossf(torch.sin(pred_p), batch[:, 1, ...].sin())
    ang_loss = cos_loss + sin_loss
    mag_loss = lossf(pred_m, batch[:, 0, ...])
    return ang_loss + mag_loss * 0.2, ang_loss, mag_loss


def check(name, a, b, tol):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().numpy() if torch.is_tensor(b) else np.asarray(b)
    err = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)
    assert err <= tol, f"oracle != reference for {name}: rel err {err:.3e} > {tol}"
    return err


def gen_case(refmodel, C, L, B, adam_steps=0):
    pn = detgen.make_params(C, seed=0)
    batch = torch.from_numpy(detgen.make_batch(B, C, L, seed=1))
    m = ref_model(refmodel, C, pn)
    out = {}
    pred = m.forward(batch[:, 0])
    loss, ang, mag = ref_loss(pred, batch, C)
    loss.backward()
    out["out"] = pred.detach().numpy()
    out["loss"] = np.array([loss.item(), ang.item(), mag.item()], np.float64)
    sd = m.model.state_dict()
    for k in detgen.param_order():
        out["grad/" + k] = dict(m.model.named_parameters())[k].grad.numpy().copy()
    for k in detgen.BN_KEYS:
        out["stat/" + k + ".running_mean"] = sd[k + ".running_mean"].numpy().copy()
        out["stat/" + k + ".running_var"] = sd[k + ".running_var"].numpy().copy()

    # --- oracle must reproduce all of it ----------------------------------------------------------
    po = unet_ref.to_torch(pn)
    for k in detgen.param_order():
        po[k].requires_grad_(True)
    stats = {k: po[k].clone() for k in po if "running" in k or "num_batches" in k}
    cap = {}
    o = unet_ref.unet_forward(po, batch[:, 0], stats, cap)
    lo, ao, mo = unet_ref.phase_loss(o, batch)
    lo.backward()
    e = check("out", o, out["out"], 2e-6)
    check("loss", torch.stack([lo, ao, mo]).double(), out["loss"], 1e-6)
    for k in detgen.param_order():
        e = max(e, check("grad " + k, po[k].grad, out["grad/" + k], 2e-5))
    for k in detgen.BN_KEYS:
        check("rm " + k, stats[k + ".running_mean"], out["stat/" + k + ".running_mean"], 1e-6)
        check("rv " + k, stats[k + ".running_var"], out["stat/" + k + ".running_var"], 1e-6)
    for k, v in cap.items():
        out["act/" + k] = v.detach().numpy().copy()

    if adam_steps:
        m = ref_model(refmodel, C, pn)
        opt = torch.optim.Adam(m.parameters(), lr=0.001)
        po = unet_ref.to_torch(pn)
        stats = {k: po[k] for k in po if "running" in k or "num_batches" in k}
        ost = unet_ref.new_opt_state({k: po[k] for k in detgen.param_order()})
        losses = []
        for s in range(adam_steps):
            b = torch.from_numpy(detgen.make_batch(B, C, L, seed=1 + s))
            opt.zero_grad()
            pr = m.forward(b[:, 0])
            ls, a_, m_ = ref_loss(pr, b, C)
            ls.backward()
            opt.step()
            losses.append([ls.item(), a_.item(), m_.item()])
            pp = {k: po[k] for k in detgen.param_order()}
            pp.update(stats)
            lo_, _, _, _ = unet_ref.train_step(pp, b, ost, stats)
            check(f"adam loss step {s}", lo_.double(), ls.item(), 1e-5)
        named = dict(m.model.named_parameters())
        for k in detgen.param_order():
            out[f"adam{adam_steps}/p/" + k] = named[k].detach().numpy().copy()
            st = opt.state[named[k]]
            out[f"adam{adam_steps}/m/" + k] = st["exp_avg"].numpy().copy()
            out[f"adam{adam_steps}/v/" + k] = st["exp_avg_sq"].numpy().copy()
            check("adam p " + k, po[k], out[f"adam{adam_steps}/p/" + k], 2e-5)
            check("adam m " + k, ost["m"][k], out[f"adam{adam_steps}/m/" + k], 2e-4)
        sd = m.model.state_dict()
        for k in detgen.BN_KEYS:
            out[f"adam{adam_steps}/stat/" + k + ".running_mean"] = sd[k + ".running_mean"].numpy().copy()
            out[f"adam{adam_steps}/stat/" + k + ".running_var"] = sd[k + ".running_var"].numpy().copy()
        out["adam_losses"] = np.array(losses, np.float64)
    print(f"  case C={C} L={L} B={B}: oracle==reference (max rel err {e:.2e}); {len(out)} arrays")
    return out


def gen_g4(refdata):
    d = detgen.normal(11, (4, 2, 16, 24))
    d[0, :, 0, 0] = 0.0                 # zero -> |z| = 0, angle 0
    d[0, 0, 0, 1], d[0, 1, 0, 1] = -1.5, 0.0   # negative real axis -> +pi
    d[0, 0, 0, 2], d[0, 1, 0, 2] = -1.5, -0.0  # -0.0 imaginary -> -pi (np.angle via arctan2)
    d[0, 0, 0, 3], d[0, 1, 0, 3] = 0.0, 2.0    # +pi/2
    d[0, 0, 0, 4], d[0, 1, 0, 4] = 0.0, -2.0   # -pi/2
    ref = refdata.get_spec_and_angle(d)
    orc = signal_ref.get_spec_and_angle(d)
    assert ref.dtype == np.float32 and ref.shape == (4, 2, 16, 24)
    check("G4", orc, ref, 1e-6)
    return {"input": d, "output": ref}


def gen_g5(refmodel):
    C, L, n_fft, hop = 16, 24, 32, 8
    pn = detgen.make_params(C, seed=0)
    clip = detgen.make_clip(hop * (L - 1), seed=2)
    spec = signal_ref.chunk_and_stft(clip, n_fft, hop)               # (2, 16, 24)
    assert spec.shape == (2, C, L), spec.shape
    polar = signal_ref.get_spec_and_angle(spec[None])[0].astype(np.float32)
    m = ref_model(refmodel, C, pn)
    with torch.no_grad():
        pred = m.forward(torch.from_numpy(polar[None, 0])).numpy()[0]
    hyb = signal_ref.hybrid_spectrum(polar[0], pred[:C])
    audio = signal_ref.generate_audio(hyb, hop, is_stft=True)
    return {"clip": clip, "spec": spec, "polar": polar, "pred": pred, "audio": audio.astype(np.float32)}


def gen_g6(refmodel):
    C, L, B = 1024, 128, 1
    print("  G6: generating 612M deterministic weights ...", flush=True)
    pn = detgen.make_params(C, seed=0)
    x = torch.from_numpy(detgen.make_batch(B, C, L, seed=1)[:, 0])
    m = ref_model(refmodel, C, pn)
    acts = {}
    print("  G6: reference forward ...", flush=True)
    with torch.no_grad():
        out = m.forward(x).numpy()
    po = unet_ref.to_torch(pn)
    cap = {}
    with torch.no_grad():
        o = unet_ref.unet_forward(po, x, None, cap)
    check("G6 out", o, out, 2e-5)
    res = {}
    for k, v in cap.items():
        v = v.numpy().astype(np.float64)
        res["stat/" + k] = np.array([v.mean(), np.abs(v).max(), np.sqrt((v * v).sum())])
    idx = (detgen._hash(99, 4096) % np.uint64(out.size)).astype(np.int64)
    res["sample_idx"] = idx
    res["sample_val"] = out.reshape(-1)[idx]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also generate G6 (full size, ~1 min, ~8 GB RAM)")
    a = ap.parse_args()
    assert os.path.isdir(REF), "reference not present: fixtures can only be generated in the build container"
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    refmodel, refdata = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    cases = [(8, 24, 1, 0), (8, 64, 3, 3), (16, 24, 3, 0), (16, 128, 2, 0), (8, 128, 3, 0), (16, 64, 1, 0)]
    for C, L, B, steps in cases:
        np.savez_compressed(os.path.join(GOLD, f"unet_C{C}_L{L}_B{B}.npz"), **gen_case(refmodel, C, L, B, steps))
    np.savez_compressed(os.path.join(GOLD, "polar_g4.npz"), **gen_g4(refdata))
    np.savez_compressed(os.path.join(GOLD, "demo_g5.npz"), **gen_g5(refmodel))
    if a.full:
        np.savez_compressed(os.path.join(GOLD, "full_g6.npz"), **gen_g6(refmodel))
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()

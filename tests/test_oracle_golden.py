"""CPU: the oracle (oracle/unet_ref.py) against the committed golden fixtures, which are outputs of the IMPORTED
reference (oracle/gen_golden.py, build container).  This is what pins the oracle everywhere the reference cannot
travel.  Tolerances are tight (the oracle and the reference run the same stock torch fp32 CPU ops)."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_ref
from phasegen import detgen

CASES = [(8, 24, 1), (8, 64, 3), (16, 24, 3), (16, 128, 2), (8, 128, 3), (16, 64, 1)]


def rel(a, b):
    a = a.detach().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.mark.parametrize("case", CASES)
def test_oracle_forward_backward_vs_golden(case, golden_dir):
    C, L, B = case
    gold = np.load(os.path.join(golden_dir, f"unet_C{C}_L{L}_B{B}.npz"))
    p = unet_ref.to_torch(detgen.make_params(C, seed=0))
    for k in detgen.param_order():
        p[k].requires_grad_(True)
    stats = {k: p[k].clone() for k in p if "running" in k or "num_batches" in k}
    batch = torch.from_numpy(detgen.make_batch(B, C, L, seed=1))
    cap = {}
    out = unet_ref.unet_forward(p, batch[:, 0], stats, cap)
    loss, ang, mag = unet_ref.phase_loss(out, batch)
    loss.backward()
    assert rel(out, gold["out"]) < 1e-5
    assert np.allclose([loss.item(), ang.item(), mag.item()], gold["loss"], rtol=1e-5)
    for k, v in cap.items():
        assert rel(v, gold["act/" + k]) < 1e-5, k
    for k in detgen.param_order():
        assert rel(p[k].grad, gold["grad/" + k]) < 1e-4, k
    for k in detgen.BN_KEYS:
        assert rel(stats[k + ".running_var"], gold["stat/" + k + ".running_var"]) < 1e-5
        assert int(stats[k + ".num_batches_tracked"]) == 1


def test_oracle_adam_vs_golden(golden_dir):
    C, L, B = 8, 64, 3
    gold = np.load(os.path.join(golden_dir, f"unet_C{C}_L{L}_B{B}.npz"))
    po = unet_ref.to_torch(detgen.make_params(C, seed=0))
    stats = {k: po[k] for k in po if "running" in k or "num_batches" in k}
    pp = {k: po[k] for k in detgen.param_order()}
    ost = unet_ref.new_opt_state(pp)
    pp.update(stats)
    for s in range(3):
        b = torch.from_numpy(detgen.make_batch(B, C, L, seed=1 + s))
        lo, ao, mo, _ = unet_ref.train_step(pp, b, ost, stats)
        assert np.allclose([lo.item(), ao.item(), mo.item()], gold["adam_losses"][s], rtol=2e-5)
    for k in detgen.param_order():
        assert rel(pp[k], gold["adam3/p/" + k]) < 1e-4, k
        assert rel(ost["v"][k], gold["adam3/v/" + k]) < 1e-3, k


def test_detgen_is_stable():
    """The deterministic generator is the contract between fixtures and every machine: pin a few values."""
    u = detgen.uniform(7, (5,), -1.0, 1.0)
    assert u.dtype == np.float32
    assert np.array_equal(u, detgen.uniform(7, (5,), -1.0, 1.0))
    big = detgen.uniform(3, ((1 << 24) + 5,), 0.0, 1.0)          # crosses the internal chunk boundary
    assert np.array_equal(big[(1 << 24) - 2:], detgen.uniform(3, ((1 << 24) + 5,), 0.0, 1.0)[(1 << 24) - 2:])
    assert np.array_equal(big[:4], detgen.uniform(3, (4,), 0.0, 1.0))
    p = detgen.make_params(8)
    assert list(p.keys())[0] == detgen.K_D0 and p[detgen.K_U0].shape == (32, 16, 32)
    assert len(detgen.state_dict_order()) == 38 and len(detgen.param_order()) == 20


def test_polar_vs_reference_golden(golden_dir):
    from oracle import signal_ref
    g = np.load(os.path.join(golden_dir, "polar_g4.npz"))
    out = signal_ref.get_spec_and_angle(g["input"])
    assert np.max(np.abs(out - g["output"])) < 1e-6
    # reference quirk (data.py:40): `d[:,0] + d[:,1]*1j` turns an imaginary part of -0.0 into +0.0, so BOTH points on
    # the negative real axis (im = +0.0 and im = -0.0) map to +pi; a bare arctan2(-0.0, -1.5) would give -pi.
    assert g["input"][0, 1, 0, 2] == 0 and np.signbit(g["input"][0, 1, 0, 2])
    assert g["output"][0, 1, 0, 1] == np.float32(np.pi) and g["output"][0, 1, 0, 2] == np.float32(np.pi)
    assert g["output"][0, 0, 0, 0] == 0 and g["output"][0, 1, 0, 0] == 0

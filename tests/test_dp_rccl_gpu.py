"""GPU: the data-parallel training step on the real backend.  Only one GPU is available to these tests, so the RCCL
("nccl") process group has a single rank: the bucketed all-reduce then is the identity, but every collective is really
issued on the gradient-arena slices from inside backward, on RCCL's stream, and waited for before Adam -- which is the
code path bench.py --gpus N and train.py take.  (Cross-rank averaging itself is covered by tests/test_dp_gloo.py.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from phasegen import detgen

pytestmark = pytest.mark.gpu


def test_trainer_step_through_rccl_single_rank_group():
    from phasegen.model import UNetModel
    from phasegen.trainer import Trainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        C, L, B = 16, 64, 2
        pn = detgen.make_params(C, seed=0)
        batch = torch.from_numpy(detgen.make_batch(B, C, L, seed=1)).cuda()
        m1 = UNetModel(C, 2 * C).load_numpy(pn)
        m2 = UNetModel(C, 2 * C).load_numpy(pn)
        t1 = Trainer(m1, always_reduce=True)          # every bucket goes through dist.all_reduce (RCCL)
        t2 = Trainer(m2)                              # plain single-process step
        assert t1.reducer.always and t1.reducer.world == 1
        for _ in range(2):
            l1 = t1.step(batch).clone()
            l2 = t2.step(batch).clone()
        assert torch.equal(l1, l2)
        assert torch.equal(m1.engine.arena.flat, m2.engine.arena.flat)
        assert torch.equal(m1.engine.arena.grad, m2.engine.arena.grad)
        # bf16 payload: the same step with every bucket rounded to bf16, reduced by RCCL as bf16 and widened back
        m3 = UNetModel(C, 2 * C).load_numpy(pn)
        t3 = Trainer(m3, always_reduce=True, grad_compress="bf16")
        l3 = t3.step(batch).clone()
        m4 = UNetModel(C, 2 * C).load_numpy(pn)
        l4 = Trainer(m4).step(batch).clone()
        assert torch.equal(l3, l4)                                      # the forward pass does not see the payload format
        g3, g4 = m3.engine.arena.grad, m4.engine.arena.grad
        assert torch.equal(g3, g4.to(torch.bfloat16).float())          # exactly the bf16 rounding of the fp32 gradient
        dist.barrier()
    finally:
        dist.destroy_process_group()

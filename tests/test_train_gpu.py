"""The flat-buffer train step on the GPU vs the CPU oracle + torch.optim.AdamW (train.py:37-53 semantics)."""
import numpy as np
import pytest
import torch

from oracle import promptir_ref as O
from promptir_amd import weights as W
from tests import util

pytestmark = pytest.mark.gpu


def test_two_train_steps_match_oracle_adamw():
    from net.model import PromptIR
    from promptir_amd.train import UNUSED_PREFIXES, DataParallelTrainer

    dev = torch.device("cuda:0")
    ctor = dict(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    net = PromptIR(**ctor)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(util.params_for(shapes, 9))
    net.to(dev)
    trainer = DataParallelTrainer(net, lr=2e-4)
    degraded, clean = W.synthetic_pair(2, 64, 64, sigma=[15, 50], seed=9)
    x, t = torch.from_numpy(degraded), torch.from_numpy(clean)

    params = {k: v.clone().requires_grad_(True) for k, v in util.params_for(shapes, 9).items()}
    opt = torch.optim.AdamW(list(params.values()), lr=2e-4)   # train.py:53
    losses_ref, losses = [], []
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss = O.l1_loss(O.promptir_forward(params, x), t)
        loss.backward()
        opt.step()
        losses_ref.append(float(loss.detach()))
        losses.append(float(trainer.train_step(x.to(dev), t.to(dev))))
    assert np.allclose(losses, losses_ref, atol=2e-6)
    sd = net.state_dict()
    worst = 0.0
    for k, ref in params.items():
        got = sd[k].cpu()
        if k.startswith(UNUSED_PREFIXES):
            assert torch.equal(got, util.params_for({k: shapes[k]}, 9)[k])   # never touched, like grad-less params in DDP
            continue
        worst = max(worst, float((got - ref.detach()).abs().max()))
    # two AdamW steps move every weight by ~lr each; agreement to 5% of one step is a tight check of grads' signs/scales
    assert worst <= 1e-5, worst
    # gradients land in the flat buffer (sinks), p.grad aliases it
    assert net.output.weight.grad.data_ptr() == trainer.opt.grad.data_ptr() + 4 * trainer.opt.offsets["output.weight"]

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


def test_graph_and_part_streams_match_the_eager_step():
    """hipGraph replay with four part-batch streams == the eager single-stream step (same kernels, the batch sum of
    the weight gradients is only re-associated across the parts)."""
    from net.model import PromptIR
    from promptir_amd.train import DataParallelTrainer

    dev = torch.device("cuda:0")
    ctor = dict(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    shapes = None
    degraded, clean = W.synthetic_pair(8, 64, 64, sigma=[15, 25, 50, 25, 15, 50, 25, 15], seed=4)
    x, t = torch.from_numpy(degraded).to(dev), torch.from_numpy(clean).to(dev)
    results = []
    for graph, streams in ((False, 1), (True, 2), (True, 1)):
        net = PromptIR(**ctor)
        shapes = shapes or {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(util.params_for(shapes, 11))
        net.to(dev)
        trainer = DataParallelTrainer(net, lr=2e-4, micro_streams=streams, graph=graph)
        losses = [float(trainer.train_step(x, t)) for _ in range(3)]
        results.append((losses, trainer.opt.grad.clone(), trainer.opt.param.clone()))
    ref_losses, ref_grad, ref_param = results[0]
    for losses, grad, param in results[1:]:
        assert np.allclose(losses, ref_losses, rtol=0, atol=2e-6), (losses, ref_losses)
        assert float((grad - ref_grad).abs().max()) <= 2e-5 * float(ref_grad.abs().max())
        assert float((param - ref_param).abs().max()) <= 2e-6

"""HIP modules vs. the reference's golden vectors AND the CPU oracle (same seeded inputs).

Tolerances (fp32 path; north_star bar is 1e-4 max-abs on the network output):
  forward  : 1e-4 * max(1, |y|max)
  gradients: 2e-4 relative to the tensor's max magnitude
"""
import json

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4
GRAD_RTOL = 2e-4


def _build(tag):
    import promptir_amd.model as M

    parts = tag.split("_")
    kind = parts[0]
    bias = tag.endswith("_bias")
    if kind == "attn":
        return M.Attention(int(parts[1]), int(parts[2]), bias)
    if kind == "ffn":
        return M.FeedForward(int(parts[1]), 2.66, bias)
    if kind == "ln":
        return M.LayerNorm(48, "WithBias" if parts[1] == "withbias" else "BiasFree")
    if kind == "tblock":
        return M.TransformerBlock(int(parts[1]), int(parts[2]), 2.66, bias,
                                  "BiasFree" if tag.endswith("biasfree") else "WithBias")
    if kind == "down":
        return M.Downsample(48)
    if kind == "up":
        return M.Upsample(96)
    if kind == "patch":
        return M.OverlapPatchEmbed(3, 48)
    if tag in ("prompt_64_id", "prompt_64_down"):
        return M.PromptGenBlock(16, 5, 16, 24)
    if tag == "prompt_aniso":
        return M.PromptGenBlock(20, 5, 8, 12)
    if tag == "prompt_up":
        return M.PromptGenBlock(8, 5, 8, 12)
    raise KeyError(tag)


def _tags():
    out = []
    for f in ("modules.npz", "modules_bias.npz", "modules_r4.npz"):   # bias=True modules: net/model.py:88-92,111-113; r4: one head of 96 channels
        z = util.load_npz(f)
        out += sorted({k.split("/")[0] for k in z.files})
    return out


@pytest.mark.parametrize("tag", _tags())
def test_module_parity(tag):
    from tests.test_oracle_golden import _module_file, _run_module

    dev = torch.device("cuda:0")
    z = util.load_npz(_module_file(tag))
    shapes = {k: tuple(v) for k, v in json.loads(str(z[f"{tag}/param_shapes"])).items()}
    y_ref = z[f"{tag}/y"]
    x_cpu, dy_cpu = util.module_inputs(tag, z[f"{tag}/x_shape"], y_ref.shape)

    mod = _build(tag)
    assert {k: tuple(v.shape) for k, v in mod.state_dict().items()} == shapes
    mod.load_state_dict(util.params_for(shapes, 7, prefix=tag + "/"))
    mod.to(dev)
    x = x_cpu.to(dev).requires_grad_(True)
    y = mod(x)
    y.backward(dy_cpu.to(dev))
    torch.cuda.synchronize()

    # 1) against the reference's golden vectors
    scale = max(1.0, float(np.abs(y_ref).max()))
    assert float((y.detach().cpu() - torch.from_numpy(y_ref)).abs().max()) <= FWD_TOL * scale
    dx_ref = z[f"{tag}/dx"]
    assert float((x.grad.cpu() - torch.from_numpy(dx_ref)).abs().max()) <= GRAD_RTOL * max(1.0, float(np.abs(dx_ref).max()))

    # 2) against the CPU oracle, full tensors for every parameter gradient
    p = util.params_for(shapes, 7, prefix=tag + "/", requires_grad=True)
    xo = x_cpu.clone().requires_grad_(True)
    yo = _run_module(tag, p, xo)
    yo.backward(dy_cpu)
    for name, prm in mod.named_parameters():
        ref = p[name].grad
        assert prm.grad is not None, name
        tol = GRAD_RTOL * max(float(ref.abs().max()), 1e-3)
        err = float((prm.grad.cpu() - ref).abs().max())
        assert err <= tol, (name, err, tol)

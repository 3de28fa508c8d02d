"""Pins the CPU oracle (oracle/promptir_ref.py) to outputs of the REAL reference.

The fixtures under tests/golden/ were produced by oracle/make_golden.py, which
imports /root/reference/net/model.py.  CPU-only; no HIP code involved.
Tolerances: both sides are fp32 PyTorch-CPU, differing only in reduction order,
so 2e-5 absolute on activations is ample (observed ~1e-6).
"""
import json

import numpy as np
import pytest
import torch

from oracle import promptir_ref as O
from tests import util

ATOL = 2e-5


def _check_grads(z, prefix, named_grads, rtol=2e-4):
    names = json.loads(str(z[prefix + "grad_names"]))
    norms = z[prefix + "grad_norm"]
    dots = z[prefix + "grad_probe"]
    scale = np.nanmax(norms)
    checked = 0
    for name, gn, gd in zip(names, norms, dots):
        g = named_grads.get(name)
        if np.isnan(gn):
            assert g is None or float(g.abs().max()) == 0.0, f"{name}: reference has no grad"
            continue
        assert g is not None, f"{name}: missing grad"
        n, d = util.grad_probe(name, g)
        assert abs(n - gn) <= rtol * max(gn, 1e-3 * scale), (name, n, gn)
        assert abs(d - gd) <= rtol * max(gn, 1e-3 * scale), (name, d, gd)
        key = f"{prefix}grad/{name}"
        if key in z.files:
            ref = z[key]
            tol = rtol * max(float(np.abs(ref).max()), 1e-3 * scale)
            assert float(np.abs(g.detach().numpy() - ref).max()) <= tol, name
        checked += 1
    return checked


MODEL_CASES = [("model_small_64.npz", True), ("model_small_72x88.npz", True), ("model_small_128.npz", False),
               ("model_full_64.npz", True), ("model_full_128.npz", False), ("model_full_128_bwd.npz", True),
               ("model_small_64_bias.npz", True)]


@pytest.mark.parametrize("fname,backward", MODEL_CASES)
def test_model_matches_reference(fname, backward):
    z = util.load_npz(fname)
    ctor = json.loads(str(z["ctor"]))
    seed = int(z["seed"])
    shapes = util.small_model_shapes(ctor.get("num_blocks", (4, 6, 6, 8)), ctor.get("num_refinement_blocks", 4),
                                     bias=ctor.get("bias", False))
    params = util.params_for(shapes, seed, requires_grad=backward)
    x = torch.from_numpy(z["x"])
    with torch.set_grad_enabled(backward):
        y = O.promptir_forward(params, x)
    assert float((y.detach() - torch.from_numpy(z["y"])).abs().max()) <= ATOL
    if backward:
        loss = O.l1_loss(y, torch.from_numpy(z["clean"]))
        assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-6
        loss.backward()
        n = _check_grads(z, "", {k: v.grad for k, v in params.items()})
        assert n > 100
        if fname == "model_full_128_bwd.npz":
            # FULL gradients of large tensors at full depth (norm + probe cannot tell equal-norm gradients apart)
            zf = util.load_npz("model_full_128_bwd_fullgrads.npz")
            assert abs(float(zf["loss"]) - float(z["loss"])) <= 1e-9
            keys = [k for k in zf.files if k.startswith("grad/")]
            assert len(keys) >= 8
            for k in keys:
                ref, g = zf[k], params[k[5:]].grad.numpy()
                assert ref.shape == g.shape
                assert float(np.abs(g - ref).max()) <= 2e-4 * float(np.abs(ref).max()), k


def _module_tags(fname="modules.npz"):
    z = util.load_npz(fname)
    return sorted({k.split("/")[0] for k in z.files})


R4_TAGS = ("attn_96_1", "tblock_96_1")   # dec1 / refinement MDTA shape: one head of 96 channels


def _module_file(tag):
    if tag in R4_TAGS:
        return "modules_r4.npz"
    return "modules_bias.npz" if tag.endswith("_bias") else "modules.npz"


def _run_module(tag, p, x):
    kind = tag.split("_")[0]
    if kind == "attn":
        return O.mdta(x, p, int(tag.split("_")[2]))
    if kind == "ffn":
        return O.gdfn(x, p)
    if kind == "ln":
        return O.layer_norm(x, p["body.weight"], p.get("body.bias"))
    if kind == "tblock":
        return O.transformer_block(x, p, int(tag.split("_")[2]))
    if kind == "down":
        return O.downsample(x, p)
    if kind == "up":
        return O.upsample(x, p)
    if kind == "patch":
        return torch.nn.functional.conv2d(x, p["proj.weight"], None, padding=1)
    if kind == "prompt":
        return O.prompt_gen(x, p)
    raise KeyError(tag)


@pytest.mark.parametrize("tag", _module_tags() + _module_tags("modules_bias.npz") + _module_tags("modules_r4.npz"))
def test_module_matches_reference(tag):
    z = util.load_npz(_module_file(tag))
    shapes = {k: tuple(v) for k, v in json.loads(str(z[f"{tag}/param_shapes"])).items()}
    p = util.params_for(shapes, 7, prefix=tag + "/", requires_grad=True)
    y_ref = z[f"{tag}/y"]
    x, dy = util.module_inputs(tag, z[f"{tag}/x_shape"], y_ref.shape)
    x.requires_grad_(True)
    y = _run_module(tag, p, x)
    assert float((y.detach() - torch.from_numpy(y_ref)).abs().max()) <= ATOL * max(1.0, float(np.abs(y_ref).max()))
    y.backward(dy)
    dx_ref = z[f"{tag}/dx"]
    assert float((x.grad - torch.from_numpy(dx_ref)).abs().max()) <= 2e-4 * max(1.0, float(np.abs(dx_ref).max()))
    _check_grads(z, tag + "/", {k: v.grad for k, v in p.items()})


def test_tile_eval_matches_reference_model():
    z = util.load_npz("tile_eval_small_160x192.npz")
    ctor = json.loads(str(z["ctor"]))
    params = util.params_for(util.small_model_shapes(ctor["num_blocks"], ctor["num_refinement_blocks"]), int(z["seed"]))
    with torch.no_grad():
        y = O.tile_eval(lambda t: O.promptir_forward(params, t), torch.from_numpy(z["x"]), 128, 32)
    assert float((y - torch.from_numpy(z["y"])).abs().max()) <= ATOL
    assert O.tile_starts(512, 128, 32) == [0, 96, 192, 288, 384]


def test_scheduler_matches_reference():
    lr = util.load_npz("scheduler_lr.npz")["lr"]
    mine = np.array([O.warmup_cosine_lr(e) for e in range(151)])
    assert np.allclose(mine, lr, rtol=1e-12, atol=1e-18)
    assert mine[0] == 0.0 and abs(mine[15] - 2e-4) < 1e-18


def test_structure_known_answers():
    shapes = util.full_shapes()
    assert len(shapes) == 548
    total = sum(int(np.prod(s)) for s in shapes.values())
    assert total == 35_592_263
    dead = [k for k in shapes if k.startswith("chnl_reduce") or k.startswith("reduce_noise_channel_")]
    assert sum(int(np.prod(shapes[k])) for k in dead) == 215_296
    hid = [O.hidden_features(c, 2.66) for c in (48, 96, 192, 384, 704, 320, 160)]
    assert hid == [127, 255, 510, 1021, 1872, 851, 425]


def test_pad_rules():
    x = torch.arange(2 * 3 * 13 * 18, dtype=torch.float32).reshape(2, 3, 13, 18)
    p, h, w = O.pad_input(x, 8)
    assert p.shape[-2:] == (16, 24) and (h, w) == (13, 18)
    p2, _, _ = O.pad_input(torch.zeros(1, 3, 16, 24), 8)
    assert p2.shape[-2:] == (16, 24)
    m, H, W = O.mirror_pad_64(torch.zeros(1, 3, 64, 100))
    assert m.shape[-2:] == (128, 128)  # test.py:101 always adds at least one row of padding


def test_psnr_known_answer():
    a = torch.full((1, 3, 4, 4), 0.5)
    b = torch.full((1, 3, 4, 4), 0.6)
    assert abs(O.psnr(a, b) - 20.0) < 1e-4  # fp32 inputs: 0.6f-0.5f is 0.1 only to ~1e-8

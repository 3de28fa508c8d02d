"""Tiled inference: GPU gather / batched model / blend vs the reference-model golden
fixture and the oracle's restatement of demo.py / test.py."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import promptir_ref as O
from promptir_amd import weights as W
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _small_net(seed):
    from net.model import PromptIR

    ctor = dict(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    net = PromptIR(**ctor)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(util.params_for(shapes, seed))
    return net.to(DEV), shapes


def test_tile_eval_vs_reference_model_golden():
    from promptir_amd.tile import tile_eval

    z = util.load_npz("tile_eval_small_160x192.npz")
    net, _ = _small_net(int(z["seed"]))
    y = tile_eval(net, torch.from_numpy(z["x"]).to(DEV), tile=128, tile_overlap=32)
    assert float((y.cpu() - torch.from_numpy(z["y"])).abs().max()) <= 1e-4


def test_padding_rules_are_bit_exact():
    from promptir_amd.tile import mirror_pad_64, pad_input, tile_starts

    x = torch.from_numpy(W.uniform01("padimg", 2 * 3 * 21 * 35).reshape(2, 3, 21, 35))
    p, h, w = pad_input(x.to(DEV), 8)
    ref, hr, wr = O.pad_input(x, 8)
    assert (h, w) == (hr, wr) == (21, 35) and torch.equal(p.cpu(), ref)
    same, _, _ = pad_input(torch.zeros(1, 3, 16, 24, device=DEV), 8)
    assert same.shape[-2:] == (16, 24)
    x2 = torch.from_numpy(W.uniform01("padimg2", 1 * 3 * 70 * 100).reshape(1, 3, 70, 100))
    m, _, _ = mirror_pad_64(x2.to(DEV))
    mref, _, _ = O.mirror_pad_64(x2)
    assert torch.equal(m.cpu(), mref) and m.shape[-2:] == (128, 128)
    assert tile_starts(512, 128, 32) == O.tile_starts(512, 128, 32) == [0, 96, 192, 288, 384]


def test_demo_flow_on_unaligned_image_vs_oracle():
    """pad_input -> tile_eval -> crop (demo.py:122-126) on a 150x171 image, against the oracle on CPU."""
    from promptir_amd.tile import pad_input, tile_eval

    net, shapes = _small_net(13)
    params = util.params_for(shapes, 13)
    deg, _ = W.synthetic_pair(1, 150, 171, sigma=25, seed=13)
    x = torch.from_numpy(deg)
    padded, h, w = pad_input(x.to(DEV))
    y = tile_eval(net, padded, tile=128, tile_overlap=32, crop=(h, w))
    with torch.no_grad():
        pr, hr, wr = O.pad_input(x)
        ref = O.tile_eval(lambda t: O.promptir_forward(params, t), pr, 128, 32)[:, :, :hr, :wr]
    assert y.shape == ref.shape == (1, 3, 150, 171)
    assert float((y.cpu() - ref).abs().max()) <= 1e-4
    assert 0.0 <= float(y.min()) and float(y.max()) <= 1.0


def test_config4_full_size_full_depth_vs_reference_golden():
    """BASELINE config 4 at its real size: 1x3x512x512, tile 128 / overlap 32 = 25 tiles, the full-depth network, the
    tiles restored as ONE batch of 25 - against the REAL reference under the demo.py harness
    (tests/golden/tile_eval_full_512.npz, oracle/make_golden.py tilefull).  The blend kernel's result also has to equal
    the host-side blend (demo.py:35-47 restated with torch ops) of the very same per-tile HIP outputs bit for bit."""
    from net.model import PromptIR
    from promptir_amd.tile import tile_eval, tile_starts

    z = util.load_npz("tile_eval_full_512.npz")
    net = PromptIR(**json.loads(str(z["ctor"])))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(util.params_for(shapes, int(z["weights_seed"])))
    net.to(DEV)
    deg, _ = W.synthetic_pair(1, 512, 512, sigma=int(z["sigma"]), seed=int(z["seed"]))
    x = torch.from_numpy(deg).to(DEV)
    seen = []

    def model(tiles):
        seen.append(tiles.shape[0])
        out = net(tiles)
        model.outs.append(out.clone())
        return out

    model.outs = []
    y = tile_eval(model, x, tile=128, tile_overlap=32)
    assert seen == [25]                                    # one launch chain over all 25 tiles
    assert float((y.cpu() - torch.from_numpy(z["y"])).abs().max()) <= 1e-4
    starts = tile_starts(512, 128, 32)
    acc, cnt = torch.zeros_like(x), torch.zeros_like(x)
    k = 0
    for i in starts:
        for j in starts:
            acc[..., i:i + 128, j:j + 128] += model.outs[0][k:k + 1]
            cnt[..., i:i + 128, j:j + 128] += 1.0
            k += 1
    assert torch.equal(y, (acc / cnt).clamp(0.0, 1.0))


def test_mirror_pad_of_an_image_that_is_exactly_one_block_high():
    """test.py:100-104 on a 64 x 80 image: (H // 64 + 1) * 64 = 128 rows, i.e. the whole flipped image is appended (the
    derain / dehaze evaluation of a 64-row crop hit a too-strict argument check here in round 4)."""
    import torch

    from promptir_amd.tile import mirror_pad_64

    x = torch.rand(1, 3, 64, 80, device="cuda:0")
    p, h, w = mirror_pad_64(x)
    ref = torch.cat([x, torch.flip(x, [2])], 2)[:, :, :128, :]
    ref = torch.cat([ref, torch.flip(ref, [3])], 3)[:, :, :, :128]
    assert (h, w) == (64, 80) and torch.equal(p, ref)

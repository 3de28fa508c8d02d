"""Shared helpers for the parity tests (fixture loading, deterministic inputs)."""
from __future__ import annotations

import json
import os
from typing import Dict

import numpy as np
import torch

from promptir_amd import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def full_shapes() -> Dict[str, tuple]:
    with open(os.path.join(GOLDEN, "state_dict_shapes.json")) as f:
        return {k: tuple(v) for k, v in json.load(f)["shapes"].items()}


def params_for(shapes: Dict[str, tuple], seed: int, prefix: str = "", device="cpu", requires_grad=False):
    out = {}
    for k, s in shapes.items():
        t = torch.from_numpy(W.make_tensor(prefix + k, tuple(s), seed)).to(device)
        if requires_grad:
            t.requires_grad_(True)
        out[k] = t
    return out


def module_inputs(tag: str, x_shape, y_shape, seed: int = 7):
    """The x / dy streams oracle/make_golden.py fed to the reference module `tag`."""
    x = (W.uniform01(tag + "#x", int(np.prod(x_shape)), seed).reshape(tuple(x_shape)) * 2 - 1).astype(np.float32)
    dy = (W.uniform01(tag + "#dy", int(np.prod(y_shape)), seed).reshape(tuple(y_shape)) * 2 - 1).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(dy)


def probe(name: str, count: int) -> np.ndarray:
    return W.uniform01(name + "#probe", count).astype(np.float64) - 0.5


def grad_probe(name: str, grad: torch.Tensor):
    g = grad.detach().double().cpu().numpy().ravel()
    return float(np.sqrt((g * g).sum())), float((g * probe(name, g.size)).sum())


def small_model_shapes(num_blocks=(1, 1, 1, 1), num_refinement_blocks=1, bias=False) -> Dict[str, tuple]:
    """Shapes of a reduced-depth network, derived from the default one by dropping block indices.  bias=True adds a
    `.bias` [Cout] after every convolution weight the reference builds with `bias=bias` (state_dict_shapes_bias.json
    lists those keys for the default depth)."""
    if bias:
        with open(os.path.join(GOLDEN, "state_dict_shapes_bias.json")) as f:
            keys = json.load(f)["keys"]
        base = small_model_shapes(num_blocks, num_refinement_blocks)
        out = {}
        for k in keys:
            if k in base:
                out[k] = base[k]
            elif k.endswith(".bias") and k[:-5] + ".weight" in base and k not in full_shapes():
                out[k] = (base[k[:-5] + ".weight"][0],)
        return out
    stages = {"encoder_level1": num_blocks[0], "encoder_level2": num_blocks[1], "encoder_level3": num_blocks[2],
              "latent": num_blocks[3], "decoder_level3": num_blocks[2], "decoder_level2": num_blocks[1],
              "decoder_level1": num_blocks[0], "refinement": num_refinement_blocks}
    out = {}
    for k, s in full_shapes().items():
        head, _, rest = k.partition(".")
        if head in stages:
            idx = int(rest.split(".", 1)[0])
            if idx >= stages[head]:
                continue
        out[k] = s
    return out

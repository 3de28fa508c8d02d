"""Deterministic, version-independent parameter / input generator.

`torch.manual_seed` streams differ between torch releases, so every parity
fixture in this repo is produced from a counter-based generator that can be
restated in any language in a few lines:

    key   = fnv1a64(utf8(name)) ^ (seed * 0x9E3779B97F4A7C15)
    z_i   = splitmix64_mix(key + (i + 1) * 0x9E3779B97F4A7C15)
    u_i   = (z_i >> 40) * 2**-24                      # float32 in [0, 1)

The value ranges follow the reference's own initialisers so that activations stay
in the regime the network was designed for:
  * conv / linear weights : U(-1/sqrt(fan_in), 1/sqrt(fan_in))  (torch default,
    net/model.py:88-92,111-113,206,223 use nn.Conv2d / nn.Linear defaults)
  * PromptGenBlock.prompt_param : U[0, 1)            (net/model.py:221, torch.rand)
  * LayerNorm weight / bias : 1 + 0.1*U(-1,1) / 0.1*U(-1,1) (reference inits 1 / 0,
    net/model.py:56-57; perturbed so that a swapped weight/bias is caught)
  * Attention.temperature : U(0.5, 1.5)              (reference inits 1, net/model.py:109)
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Mapping, Sequence, Tuple

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for byte in name.encode("utf-8"):
        h ^= byte
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def uniform01(name: str, count: int, seed: int = 0) -> np.ndarray:
    """`count` float32 values in [0,1) for the stream called `name`."""
    key = np.uint64(fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF))
    with np.errstate(over="ignore"):
        x = key + (np.arange(1, count + 1, dtype=np.uint64) * _GOLDEN)
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        x = x ^ (x >> np.uint64(31))
    return ((x >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def normal01(name: str, count: int, seed: int = 0) -> np.ndarray:
    """Standard normal float32 values (Box-Muller over two uniform streams)."""
    u1 = uniform01(name + "#bm1", count, seed).astype(np.float64)
    u2 = uniform01(name + "#bm2", count, seed).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * math.pi * u2)).astype(np.float32)


def _role_range(name: str, shape: Sequence[int]) -> Tuple[float, float]:
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "prompt_param":
        return 0.0, 1.0
    if leaf == "temperature":
        return 0.5, 1.5
    if leaf == "weight" and len(shape) == 1:  # LayerNorm gain (the only 1-D weights)
        return 0.9, 1.1
    if leaf == "bias":  # LayerNorm shift, PromptGenBlock.linear_layer.bias
        return -0.1, 0.1
    fan_in = 1
    for d in shape[1:]:
        fan_in *= int(d)
    bound = 1.0 / math.sqrt(max(fan_in, 1))
    return -bound, bound


def make_tensor(name: str, shape: Sequence[int], seed: int = 0) -> np.ndarray:
    count = int(np.prod(shape)) if len(shape) else 1
    lo, hi = _role_range(name, shape)
    u = uniform01(name, count, seed)
    return (np.float32(lo) + u * np.float32(hi - lo)).reshape(tuple(shape))


def make_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0) -> Dict[str, np.ndarray]:
    """Values for every (name -> shape) entry; order-independent."""
    return {k: make_tensor(k, tuple(v), seed) for k, v in shapes.items()}


def synthetic_clean(batch: int, height: int, width: int, seed: int = 0, channels: int = 3) -> np.ndarray:
    """Smooth-ish clean image batch in [0,1]: 4x4 box-filtered uniform noise."""
    pad = 3
    u = uniform01("clean", batch * channels * (height + pad) * (width + pad), seed)
    u = u.reshape(batch, channels, height + pad, width + pad).astype(np.float64)
    c = np.cumsum(np.cumsum(u, axis=2), axis=3)
    c = np.pad(c, ((0, 0), (0, 0), (1, 0), (1, 0)))
    k = pad + 1
    box = c[:, :, k:, k:] - c[:, :, :-k, k:] - c[:, :, k:, :-k] + c[:, :, :-k, :-k]
    img = box / float(k * k)
    # stretch contrast back towards the full range
    img = np.clip((img - 0.5) * 3.0 + 0.5, 0.0, 1.0)
    return img.astype(np.float32)


def degrade_gaussian(clean: np.ndarray, sigma: float, seed: int = 0) -> np.ndarray:
    """sigma-noise in the uint8 domain, clip, truncate to uint8, back to [0,1].

    Mirrors the reference's test-time synthesis utils/dataset_utils.py:195-198
    (np.clip(clean + noise*sigma, 0, 255).astype(np.uint8)) followed by ToTensor.
    """
    img255 = np.floor(clean.astype(np.float64) * 255.0)  # the reference reads uint8 images
    noise = normal01("noise", clean.size, seed).reshape(clean.shape).astype(np.float64)
    noisy = np.clip(img255 + noise * float(sigma), 0, 255).astype(np.uint8)
    return noisy.astype(np.float32) / np.float32(255.0)


def synthetic_pair(batch: int, height: int, width: int, sigma=25, seed: int = 0):
    """(degraded, clean) float32 NCHW in [0,1]; sigma may be a scalar or per-sample list."""
    clean = synthetic_clean(batch, height, width, seed)
    clean = np.floor(clean * 255.0).astype(np.float32) / np.float32(255.0)
    if np.isscalar(sigma):
        degraded = degrade_gaussian(clean, float(sigma), seed)
    else:
        parts = [degrade_gaussian(clean[i:i + 1], float(s), seed + 1000 * (i + 1)) for i, s in enumerate(sigma)]
        degraded = np.concatenate(parts, axis=0)
    return degraded, clean

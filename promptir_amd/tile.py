"""Tiled inference and evaluation-time padding (reference demo.py:17-48, test.py:100-104) on the GPU.

`pad_input` / `tile_eval` keep the reference's names, arguments and results; the tiles are gathered by one
kernel, restored by the network as ONE batch (optionally chunked) and blended / clamped / cropped by one
kernel.  No op of PromptIR mixes batch entries, so this equals the reference's sequential tile loop.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch

from . import ops
from .ops import _bs, _planes, _require_gpu, _stream, check, lib


def tile_starts(extent: int, tile: int, overlap: int) -> List[int]:
    """demo.py:31-33 — list(range(0, extent-tile, stride)) + [extent-tile]."""
    stride = tile - overlap
    return list(range(0, extent - tile, stride)) + [extent - tile]


def padded_size(height: int, width: int, multiple: int = 8) -> Tuple[int, int]:
    """demo.py:18-21 — next multiple (only when not already one)."""
    H = ((height + multiple) // multiple) * multiple if height % multiple else height
    W = ((width + multiple) // multiple) * multiple if width % multiple else width
    return H, W


def _gather(x: torch.Tensor, Hp: int, Wp: int, th: int, tw: int, sh: int, sw: int, nth: int, ntw: int, mode: int):
    x = _planes(x)
    b, c, h, w = x.shape
    out = torch.empty((b * nth * ntw, c, th, tw), dtype=torch.float32, device=x.device)
    check(lib.pir_tiles_gather(x.data_ptr(), _bs(x), out.data_ptr(), b, c, h, w, Hp, Wp, th, tw, sh, sw, nth, ntw, mode,
                               _stream()), "pir_tiles_gather")
    return out


def pad_input(input_: torch.Tensor, img_multiple_of: int = 8):
    """demo.py:17-24: reflect-pad bottom/right to a multiple; returns (padded, height, width)."""
    _require_gpu(input_)
    h, w = input_.shape[2], input_.shape[3]
    Hp, Wp = padded_size(h, w, img_multiple_of)
    if (Hp, Wp) == (h, w):
        return input_, h, w
    return _gather(input_, Hp, Wp, Hp, Wp, 1, 1, 1, 1, 0), h, w


def mirror_pad_64(x: torch.Tensor):
    """test.py:100-104: extend by the flipped image up to (H//64+1)*64 (always at least one row/column)."""
    _require_gpu(x)
    h, w = x.shape[2], x.shape[3]
    Hp, Wp = (h // 64 + 1) * 64, (w // 64 + 1) * 64
    if Hp - h > h or Wp - w > w:
        raise RuntimeError("mirror padding larger than the image (the reference would silently truncate)")
    return _gather(x, Hp, Wp, Hp, Wp, 1, 1, 1, 1, 1), h, w


def tile_eval(model: Callable[[torch.Tensor], torch.Tensor], input_: torch.Tensor, tile: int = 128, tile_overlap: int = 32,
              crop: Optional[Tuple[int, int]] = None, max_batch: int = 64) -> torch.Tensor:
    """demo.py:26-48.  `crop=(h, w)` additionally applies the caller's `[:, :, :h, :w]` (demo.py:126)."""
    _require_gpu(input_)
    b, c, h, w = input_.shape
    tile = min(tile, h, w)
    assert tile % 8 == 0, "tile size should be multiple of 8"
    stride = tile - tile_overlap
    nth, ntw = len(tile_starts(h, tile, tile_overlap)), len(tile_starts(w, tile, tile_overlap))
    tiles = _gather(input_, h, w, tile, tile, stride, stride, nth, ntw, 0)
    restored = torch.empty_like(tiles)
    with torch.no_grad():
        for s in range(0, tiles.shape[0], max_batch):
            out = model(tiles[s:s + max_batch])
            ops.copy_planes(out, restored[s:s + max_batch])
    ho, wo = crop if crop is not None else (h, w)
    result = torch.empty((b, c, ho, wo), dtype=torch.float32, device=input_.device)
    check(lib.pir_tiles_blend(restored.data_ptr(), result.data_ptr(), c * ho * wo, b, c, h, w, tile, tile, stride, stride,
                              nth, ntw, ho, wo, 1, _stream()), "pir_tiles_blend")
    return result


def psnr(restored: torch.Tensor, clean: torch.Tensor) -> float:
    """utils/val_utils.py:49-62: clip both to [0,1], PSNR with data_range 1, averaged over the batch.
    Evaluated on the host like the reference does (after the device-to-host copy)."""
    import math

    r = restored.detach().double().cpu().clamp(0, 1)
    c = clean.detach().double().cpu().clamp(0, 1)
    vals = []
    for i in range(r.shape[0]):
        mse = float(((r[i] - c[i]) ** 2).mean())
        vals.append(10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf"))
    return sum(vals) / len(vals)

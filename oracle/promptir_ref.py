"""ORACLE — test infrastructure, not product code.

A CPU restatement (plain PyTorch fp32 tensor ops, functional style, parameters
passed as a flat ``{name: tensor}`` mapping with the reference's state_dict
names) of the reference's `net/model.py` forward path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s ``cpu_baseline`` leg may import
this package; `promptir_amd` never does.

Parity pin: `oracle/make_golden.py` imports the real reference
(`/root/reference/net/model.py`) in the build container, loads the repo's
deterministic weights into it and stores inputs / outputs / gradient summaries
under `tests/golden/`.  `tests/test_oracle_golden.py` checks this restatement
against those fixtures, so the oracle is PINNED to reference outputs.

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence

import torch
import torch.nn.functional as F

Params = Mapping[str, torch.Tensor]


# --------------------------------------------------------------------------- helpers
def _sub(params: Params, prefix: str) -> Dict[str, torch.Tensor]:
    """Entries of `params` below `prefix.` with the prefix removed."""
    if not prefix:
        return dict(params)
    p = prefix + "."
    return {k[len(p):]: v for k, v in params.items() if k.startswith(p)}


def hidden_features(dim: int, ffn_expansion_factor: float) -> int:
    """net/model.py:86 — int(dim*ffn_expansion_factor)."""
    return int(dim * ffn_expansion_factor)


# --------------------------------------------------------------------------- LayerNorm
def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """Per-pixel normalisation over the channel axis of an NCHW tensor.

    net/model.py:47-76 (+ to_3d/to_4d :21-25): the reference moves C last,
    takes mean / biased variance over it, eps=1e-5 inside the sqrt.
    `bias is None` selects the BiasFree variant (net/model.py:27-41), which
    does NOT subtract the mean in the numerator but still divides by the
    sqrt of the (mean-centred) variance.
    """
    mu = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, keepdim=True, unbiased=False)
    w = weight.view(1, -1, 1, 1)
    if bias is None:
        return x / torch.sqrt(var + 1e-5) * w
    return (x - mu) / torch.sqrt(var + 1e-5) * w + bias.view(1, -1, 1, 1)


def _norm(x: torch.Tensor, p: Params) -> torch.Tensor:
    return layer_norm(x, p["body.weight"], p.get("body.bias"))


# --------------------------------------------------------------------------- GDFN
def gdfn(x: torch.Tensor, p: Params) -> torch.Tensor:
    """Gated depthwise feed-forward, net/model.py:94-99.

    1x1 (C->2*hid) -> depthwise 3x3 -> split halves -> gelu(erf)(x1)*x2 -> 1x1 (hid->C).
    """
    t = F.conv2d(x, p["project_in.weight"], p.get("project_in.bias"))
    t = F.conv2d(t, p["dwconv.weight"], p.get("dwconv.bias"), padding=1, groups=t.shape[1])
    hid = t.shape[1] // 2
    gated = F.gelu(t[:, :hid]) * t[:, hid:]
    return F.conv2d(gated, p["project_out.weight"], p.get("project_out.bias"))


# --------------------------------------------------------------------------- MDTA
def mdta(x: torch.Tensor, p: Params, num_heads: int) -> torch.Tensor:
    """Multi-DConv head transposed attention, net/model.py:117-138.

    qkv = dw3x3(1x1(x)); per head q,k,v are [c, HW]; q,k L2-normalised over HW
    (F.normalize, eps 1e-12); attn = softmax(q k^T * temperature) over the last
    axis (c x c per head); out = attn v; 1x1 projection.
    """
    b, c, h, w = x.shape
    t = F.conv2d(x, p["qkv.weight"], p.get("qkv.bias"))
    t = F.conv2d(t, p["qkv_dwconv.weight"], p.get("qkv_dwconv.bias"), padding=1, groups=3 * c)
    q, k, v = (s.reshape(b, num_heads, c // num_heads, h * w) for s in t.split(c, dim=1))
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    logits = torch.matmul(q, k.transpose(-1, -2)) * p["temperature"].view(1, num_heads, 1, 1)
    out = torch.matmul(torch.softmax(logits, dim=-1), v).reshape(b, c, h, w)
    return F.conv2d(out, p["project_out.weight"], p.get("project_out.bias"))


# --------------------------------------------------------------------------- TransformerBlock
def transformer_block(x: torch.Tensor, p: Params, num_heads: int) -> torch.Tensor:
    """net/model.py:192-196 — two pre-norm residual branches."""
    x = x + mdta(_norm(x, _sub(p, "norm1")), _sub(p, "attn"), num_heads)
    x = x + gdfn(_norm(x, _sub(p, "norm2")), _sub(p, "ffn"))
    return x


def _stage(x: torch.Tensor, params: Params, prefix: str, num_heads: int) -> torch.Tensor:
    """An nn.Sequential of TransformerBlocks (net/model.py:278,283,288,291,299,308,316,318)."""
    idx = 0
    while f"{prefix}.{idx}.norm1.body.weight" in params:
        x = transformer_block(x, _sub(params, f"{prefix}.{idx}"), num_heads)
        idx += 1
    return x


# --------------------------------------------------------------------------- resampling
def downsample(x: torch.Tensor, p: Params) -> torch.Tensor:
    """net/model.py:160-168 — dense 3x3 (C->C/2) then PixelUnshuffle(2)."""
    return F.pixel_unshuffle(F.conv2d(x, p["body.0.weight"], None, padding=1), 2)


def upsample(x: torch.Tensor, p: Params) -> torch.Tensor:
    """net/model.py:170-178 — dense 3x3 (C->2C) then PixelShuffle(2)."""
    return F.pixel_shuffle(F.conv2d(x, p["body.0.weight"], None, padding=1), 2)


# --------------------------------------------------------------------------- PromptGenBlock
def prompt_gen(x: torch.Tensor, p: Params) -> torch.Tensor:
    """net/model.py:226-235.

    emb = spatial mean; weights = softmax(Linear(emb)); prompt = sum_l w_l * P_l;
    bilinear resize (align_corners=False) to the feature size; dense 3x3.
    """
    b, _, h, w = x.shape
    emb = x.mean(dim=(2, 3))
    mix = torch.softmax(F.linear(emb, p["linear_layer.weight"], p["linear_layer.bias"]), dim=1)
    prompts = p["prompt_param"][0]  # [L, D, S, S]
    prompt = torch.einsum("bl,ldst->bdst", mix, prompts)
    prompt = F.interpolate(prompt, size=(h, w), mode="bilinear", align_corners=False)
    return F.conv2d(prompt, p["conv3x3.weight"], None, padding=1)


# --------------------------------------------------------------------------- whole network
def promptir_forward(params: Params, inp_img: torch.Tensor, heads: Sequence[int] = (1, 2, 4, 8),
                     decoder: bool = True) -> torch.Tensor:
    """net/model.py:322-380.  `heads[2]` is used for all three noise_level blocks
    (net/model.py:295,304,312).  The `.bias` entries exist only for PromptIR(bias=True) (:253; patch_embed, the
    Down/Upsample and PromptGenBlock convolutions are always bias-free, :164,174,206,223,259)."""
    P = params
    enc1_in = F.conv2d(inp_img, P["patch_embed.proj.weight"], None, padding=1)        # :324
    enc1 = _stage(enc1_in, P, "encoder_level1", heads[0])                              # :326
    enc2 = _stage(downsample(enc1, _sub(P, "down1_2")), P, "encoder_level2", heads[1])  # :328-330
    enc3 = _stage(downsample(enc2, _sub(P, "down2_3")), P, "encoder_level3", heads[2])  # :332-334
    latent = _stage(downsample(enc3, _sub(P, "down3_4")), P, "latent", heads[3])        # :336-337
    if decoder:                                                                        # :339-343
        latent = torch.cat([latent, prompt_gen(latent, _sub(P, "prompt3"))], 1)
        latent = transformer_block(latent, _sub(P, "noise_level3"), heads[2])
        latent = F.conv2d(latent, P["reduce_noise_level3.weight"], P.get("reduce_noise_level3.bias"))
    d3 = torch.cat([upsample(latent, _sub(P, "up4_3")), enc3], 1)                       # :346-347
    d3 = F.conv2d(d3, P["reduce_chan_level3.weight"], P.get("reduce_chan_level3.bias"))                                   # :348
    d3 = _stage(d3, P, "decoder_level3", heads[2])                                      # :350
    if decoder:                                                                        # :351-355
        d3 = torch.cat([d3, prompt_gen(d3, _sub(P, "prompt2"))], 1)
        d3 = transformer_block(d3, _sub(P, "noise_level2"), heads[2])
        d3 = F.conv2d(d3, P["reduce_noise_level2.weight"], P.get("reduce_noise_level2.bias"))
    d2 = torch.cat([upsample(d3, _sub(P, "up3_2")), enc2], 1)                           # :358-359
    d2 = F.conv2d(d2, P["reduce_chan_level2.weight"], P.get("reduce_chan_level2.bias"))                                   # :360
    d2 = _stage(d2, P, "decoder_level2", heads[1])                                      # :362
    if decoder:                                                                        # :363-367
        d2 = torch.cat([d2, prompt_gen(d2, _sub(P, "prompt1"))], 1)
        d2 = transformer_block(d2, _sub(P, "noise_level1"), heads[2])
        d2 = F.conv2d(d2, P["reduce_noise_level1.weight"], P.get("reduce_noise_level1.bias"))
    d1 = torch.cat([upsample(d2, _sub(P, "up2_1")), enc1], 1)                           # :369-370
    d1 = _stage(d1, P, "decoder_level1", heads[0])                                      # :372
    d1 = _stage(d1, P, "refinement", heads[0])                                          # :374
    return F.conv2d(d1, P["output.weight"], P.get("output.bias"), padding=1) + inp_img                 # :377


def l1_loss(restored: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
    """train.py:32,43 — nn.L1Loss() (mean absolute error)."""
    return (restored - clean).abs().mean()


# --------------------------------------------------------------------------- callers
def pad_input(x: torch.Tensor, multiple: int = 8):
    """demo.py:17-24 — reflect-pad bottom/right up to the next multiple."""
    height, width = x.shape[2], x.shape[3]
    H = ((height + multiple) // multiple) * multiple
    W = ((width + multiple) // multiple) * multiple
    padh = H - height if height % multiple != 0 else 0
    padw = W - width if width % multiple != 0 else 0
    return F.pad(x, (0, padw, 0, padh), mode="reflect"), height, width


def tile_starts(extent: int, tile: int, overlap: int) -> List[int]:
    """demo.py:31-33 — list(range(0, extent-tile, stride)) + [extent-tile]."""
    stride = tile - overlap
    return list(range(0, extent - tile, stride)) + [extent - tile]


def tile_eval(model_fn, x: torch.Tensor, tile: int = 128, tile_overlap: int = 32) -> torch.Tensor:
    """demo.py:26-48 — overlapping tiles, accumulate outputs and hit counts, divide, clamp."""
    b, c, h, w = x.shape
    tile = min(tile, h, w)
    assert tile % 8 == 0, "tile size should be multiple of 8"
    acc = torch.zeros_like(x)
    cnt = torch.zeros_like(x)
    for i in tile_starts(h, tile, tile_overlap):
        for j in tile_starts(w, tile, tile_overlap):
            acc[..., i:i + tile, j:j + tile] += model_fn(x[..., i:i + tile, j:j + tile])
            cnt[..., i:i + tile, j:j + tile] += 1.0
    return (acc / cnt).clamp(0.0, 1.0)


def mirror_pad_64(x: torch.Tensor):
    """test.py:100-104 — extend by flipped copy up to (H//64+1)*64 (always pads >= 1)."""
    _, _, H, W = x.shape
    hp = (H // 64 + 1) * 64 - H
    wp = (W // 64 + 1) * 64 - W
    x = torch.cat([x, torch.flip(x, [2])], 2)[:, :, :H + hp, :]
    x = torch.cat([x, torch.flip(x, [3])], 3)[:, :, :, :W + wp]
    return x, H, W


def psnr(restored: torch.Tensor, clean: torch.Tensor) -> float:
    """utils/val_utils.py:49-62 — clip both to [0,1]; skimage PSNR with data_range=1,
    averaged over the batch (10*log10(1/mse) per image)."""
    r = restored.detach().double().clamp(0, 1)
    c = clean.detach().double().clamp(0, 1)
    vals = []
    for i in range(r.shape[0]):
        mse = float(((r[i] - c[i]) ** 2).mean())
        vals.append(10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf"))
    return sum(vals) / len(vals)


def warmup_cosine_lr(epoch: int, base_lr: float = 2e-4, warmup_epochs: int = 15, max_epochs: int = 150,
                     warmup_start_lr: float = 0.0, eta_min: float = 0.0) -> float:
    """utils/schedulers.py:332-346 — closed form used when step(epoch) is called
    (train.py:48-50 passes current_epoch, so this is the form that is live)."""
    if epoch < warmup_epochs:
        return warmup_start_lr + epoch * (base_lr - warmup_start_lr) / (warmup_epochs - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (
        1 + math.cos(math.pi * (epoch - warmup_epochs) / (max_epochs - warmup_epochs)))

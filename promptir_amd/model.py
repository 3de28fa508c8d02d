"""PromptIR nn.Module surface over the HIP kernels.

Mirrors the reference's module tree (class names, constructor arguments, attribute
names and therefore every state_dict key / shape of /root/reference/net/model.py) so
that train / test / demo drivers and Lightning `net.*` checkpoints interchange, while
every forward/backward op dispatches to libpromptir_hip.so through `promptir_amd.ops`.

The module only runs on ROCm tensors.  A CPU tensor raises (no eager fallback); the CPU
reference lives in `oracle/` and is test infrastructure.
"""
from __future__ import annotations

import math
import numbers
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import ops


def _check_channels(x: torch.Tensor, expected: int, who: str) -> None:
    if x.dim() != 4 or x.shape[1] != expected:
        raise RuntimeError(f"{who}: expected input[B, {expected}, H, W] but got {list(x.shape)}")


class _ConvParams(nn.Module):
    """Convolution parameter holder with nn.Conv2d's parameter names, shapes, registration order and default init.
    `bias=True` (never used by the reference's callers, net/model.py:253, train.py:31) adds the bias with its own
    in-place kernel after the bias-free convolution kernel (ops.BiasAddFn)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, groups: int = 1, bias: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.groups = in_channels, out_channels, kernel_size, groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kernel_size, kernel_size))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # nn.Conv2d.reset_parameters
        if bias:
            fan_in = (in_channels // groups) * kernel_size * kernel_size
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            self.bias = nn.Parameter(torch.empty(out_channels))
            nn.init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter("bias", None)

    def _add_bias(self, y):
        return y if self.bias is None else ops.BiasAddFn.apply(y, self.bias)


class Conv1x1(_ConvParams):
    """nn.Conv2d(cin, cout, kernel_size=1, bias=False) -> fp32 MFMA GEMM (pir_gemm_nn)."""

    def __init__(self, in_channels, out_channels, bias=False):
        super().__init__(in_channels, out_channels, 1, 1, bias)

    def forward(self, x, residual: Optional[torch.Tensor] = None):
        _check_channels(x, self.in_channels, "Conv1x1")
        return self._add_bias(ops.Conv1x1Fn.apply(x, self.weight, residual))


class Conv3x3(_ConvParams):
    """nn.Conv2d(cin, cout, 3, stride=1, padding=1, bias=False) -> 9 shifted MFMA GEMMs (pir_conv3x3)."""

    def __init__(self, in_channels, out_channels, bias=False):
        super().__init__(in_channels, out_channels, 3, 1, bias)

    def forward(self, x, residual: Optional[torch.Tensor] = None, out: Optional["ops.OutSlot"] = None):
        _check_channels(x, self.in_channels, "Conv3x3")
        return self._add_bias(ops.Conv3x3Fn.apply(x, self.weight, residual, out))


class DepthwiseConv3x3(_ConvParams):
    """nn.Conv2d(C, C, 3, padding=1, groups=C, bias=False) -> LDS-tiled stencil (pir_dwconv3x3)."""

    def __init__(self, channels, bias=False):
        super().__init__(channels, channels, 3, channels, bias)

    def forward(self, x):
        _check_channels(x, self.in_channels, "DepthwiseConv3x3")
        return self._add_bias(ops.DwConvFn.apply(x, self.weight))


class PixelUnshuffle2(nn.Module):
    def forward(self, x):
        return ops.PixelUnshuffleFn.apply(x)


class PixelShuffle2(nn.Module):
    def forward(self, x, out: Optional["ops.OutSlot"] = None):
        return ops.PixelShuffleFn.apply(x, out)


##########################################################################
## Layer Norm (reference net/model.py:27-76)
class BiasFree_LayerNorm(nn.Module):
    def __init__(self, normalized_shape):
        super().__init__()
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        assert len(normalized_shape) == 1
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.normalized_shape = torch.Size(normalized_shape)

    def forward(self, x):  # x: NCHW (the to_3d/to_4d permutes are never materialised)
        return ops.LayerNormFn.apply(x, self.weight, None)


class WithBias_LayerNorm(nn.Module):
    def __init__(self, normalized_shape):
        super().__init__()
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        assert len(normalized_shape) == 1
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.normalized_shape = torch.Size(normalized_shape)

    def forward(self, x):
        return ops.LayerNormFn.apply(x, self.weight, self.bias)


class LayerNorm(nn.Module):
    def __init__(self, dim, LayerNorm_type):
        super().__init__()
        self.body = BiasFree_LayerNorm(dim) if LayerNorm_type == 'BiasFree' else WithBias_LayerNorm(dim)

    def forward(self, x):
        _check_channels(x, self.body.normalized_shape[0], "LayerNorm")
        return self.body(x)


##########################################################################
## Gated-Dconv Feed-Forward Network (reference net/model.py:82-99)
class FeedForward(nn.Module):
    def __init__(self, dim, ffn_expansion_factor, bias):
        super().__init__()
        hidden_features = int(dim * ffn_expansion_factor)
        self.project_in = Conv1x1(dim, hidden_features * 2, bias=bias)
        self.dwconv = DepthwiseConv3x3(hidden_features * 2, bias=bias)
        self.project_out = Conv1x1(hidden_features, dim, bias=bias)

    def forward(self, x, residual: Optional[torch.Tensor] = None):
        x = self.project_in(x)
        if self.dwconv.bias is None:
            x = ops.DwConvGateFn.apply(x, self.dwconv.weight)   # dwconv + chunk + gelu(x1)*x2 fused
        else:                                                   # the bias sits between the stencil and the gate
            x = ops.GeluGateFn.apply(self.dwconv(x))
        return self.project_out(x, residual)


##########################################################################
## Multi-DConv Head Transposed Self-Attention (reference net/model.py:105-138)
class Attention(nn.Module):
    def __init__(self, dim, num_heads, bias):
        super().__init__()
        if dim % num_heads:
            raise ValueError("dim must be divisible by num_heads")
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = Conv1x1(dim, dim * 3, bias=bias)
        self.qkv_dwconv = DepthwiseConv3x3(dim * 3, bias=bias)
        self.project_out = Conv1x1(dim, dim, bias=bias)

    def forward(self, x, residual: Optional[torch.Tensor] = None):
        qkv = self.qkv_dwconv(self.qkv(x))
        out = ops.MdtaCoreFn.apply(qkv, self.temperature, self.num_heads)
        return self.project_out(out, residual)


##########################################################################
## Resizing modules (reference net/model.py:160-178)
class Downsample(nn.Module):
    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(Conv3x3(n_feat, n_feat // 2), PixelUnshuffle2())

    def forward(self, x):
        return self.body(x)


class Upsample(nn.Module):
    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(Conv3x3(n_feat, n_feat * 2), PixelShuffle2())

    def forward(self, x, out: Optional["ops.OutSlot"] = None):
        return self.body[1](self.body[0](x), out)


##########################################################################
## Transformer Block (reference net/model.py:183-196)
class TransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        self.norm1 = LayerNorm(dim, LayerNorm_type)
        self.attn = Attention(dim, num_heads, bias)
        self.norm2 = LayerNorm(dim, LayerNorm_type)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)

    def forward(self, x, out: Optional["ops.OutSlot"] = None):
        _check_channels(x, self.norm1.body.normalized_shape[0], "TransformerBlock")
        a, f = self.attn, self.ffn
        if a.qkv.bias is not None:   # bias=True: composed from the per-op nodes (each conv followed by its bias add)
            x = a(self.norm1(x), residual=x)
            return f(self.norm2(x), residual=x)   # (`out` unused: the concat then copies, ops.CatChannelsFn)
        # one autograd node per block; residual adds live in the project_out epilogues (forward) and in
        # the LayerNorm-backward kernel (backward).  Equivalent to
        #   x = self.attn(self.norm1(x), residual=x); x = self.ffn(self.norm2(x), residual=x)
        return ops.TransformerBlockFn.apply(
            x, self.norm1.body.weight, getattr(self.norm1.body, "bias", None), a.temperature, a.qkv.weight,
            a.qkv_dwconv.weight, a.project_out.weight, self.norm2.body.weight, getattr(self.norm2.body, "bias", None),
            f.project_in.weight, f.dwconv.weight, f.project_out.weight, a.num_heads, not torch.is_grad_enabled(), out)


##########################################################################
## Overlapped image patch embedding with 3x3 Conv (reference net/model.py:202-211)
class OverlapPatchEmbed(nn.Module):
    def __init__(self, in_c=3, embed_dim=48, bias=False):
        super().__init__()
        self.proj = Conv3x3(in_c, embed_dim, bias=bias)

    def forward(self, x):
        return self.proj(x)


class _LinearParams(nn.Module):
    """nn.Linear parameter holder (weight [out,in], bias [out]) with nn.Linear's default init."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)


##########################################################################
## Prompt Gen Module (reference net/model.py:218-235)
class PromptGenBlock(nn.Module):
    def __init__(self, prompt_dim=128, prompt_len=5, prompt_size=96, lin_dim=192):
        super().__init__()
        self.prompt_param = nn.Parameter(torch.rand(1, prompt_len, prompt_dim, prompt_size, prompt_size))
        self.linear_layer = _LinearParams(lin_dim, prompt_len)
        self.conv3x3 = Conv3x3(prompt_dim, prompt_dim)

    def forward(self, x, out: Optional["ops.OutSlot"] = None):
        _check_channels(x, self.linear_layer.in_features, "PromptGenBlock")
        prompt = ops.PromptGenFn.apply(x, self.prompt_param, self.linear_layer.weight, self.linear_layer.bias)
        return self.conv3x3(prompt, out=out)


def _cat(a, b):
    return ops.CatChannelsFn.apply(a, b)


def _run_stage(stage: nn.Sequential, x, out: Optional["ops.OutSlot"] = None):
    """nn.Sequential of TransformerBlocks; the last block may write its output into a concat buffer slot."""
    n = len(stage)
    for i, blk in enumerate(stage):
        x = blk(x, out if i == n - 1 else None)
    return x


def _slot(buf, c0, c):
    return ops.OutSlot(buf, c0, c) if buf is not None else None


##########################################################################
## PromptIR (reference net/model.py:244-380)
class PromptIR(nn.Module):
    def __init__(self,
                 inp_channels=3,
                 out_channels=3,
                 dim=48,
                 num_blocks=[4, 6, 6, 8],
                 num_refinement_blocks=4,
                 heads=[1, 2, 4, 8],
                 ffn_expansion_factor=2.66,
                 bias=False,
                 LayerNorm_type='WithBias',
                 decoder=False,
                 ):
        super().__init__()

        def stage(width, n_heads, count):
            return nn.Sequential(*[TransformerBlock(dim=width, num_heads=n_heads,
                                                    ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                                    LayerNorm_type=LayerNorm_type) for _ in range(count)])

        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.decoder = decoder
        if self.decoder:
            self.prompt1 = PromptGenBlock(prompt_dim=64, prompt_len=5, prompt_size=64, lin_dim=96)
            self.prompt2 = PromptGenBlock(prompt_dim=128, prompt_len=5, prompt_size=32, lin_dim=192)
            self.prompt3 = PromptGenBlock(prompt_dim=320, prompt_len=5, prompt_size=16, lin_dim=384)

        # present in the reference's state_dict but never used by its forward (net/model.py:271-273,277,282,287)
        self.chnl_reduce1 = Conv1x1(64, 64, bias=bias)
        self.chnl_reduce2 = Conv1x1(128, 128, bias=bias)
        self.chnl_reduce3 = Conv1x1(320, 256, bias=bias)
        self.reduce_noise_channel_1 = Conv1x1(dim + 64, dim, bias=bias)
        self.encoder_level1 = stage(dim, heads[0], num_blocks[0])
        self.down1_2 = Downsample(dim)
        self.reduce_noise_channel_2 = Conv1x1(int(dim * 2 ** 1) + 128, int(dim * 2 ** 1), bias=bias)
        self.encoder_level2 = stage(int(dim * 2 ** 1), heads[1], num_blocks[1])
        self.down2_3 = Downsample(int(dim * 2 ** 1))
        self.reduce_noise_channel_3 = Conv1x1(int(dim * 2 ** 2) + 256, int(dim * 2 ** 2), bias=bias)
        self.encoder_level3 = stage(int(dim * 2 ** 2), heads[2], num_blocks[2])
        self.down3_4 = Downsample(int(dim * 2 ** 2))
        self.latent = stage(int(dim * 2 ** 3), heads[3], num_blocks[3])

        self.up4_3 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level3 = Conv1x1(int(dim * 2 ** 1) + 192, int(dim * 2 ** 2), bias=bias)
        self.noise_level3 = TransformerBlock(dim=int(dim * 2 ** 2) + 512, num_heads=heads[2],
                                             ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                             LayerNorm_type=LayerNorm_type)
        self.reduce_noise_level3 = Conv1x1(int(dim * 2 ** 2) + 512, int(dim * 2 ** 2), bias=bias)
        self.decoder_level3 = stage(int(dim * 2 ** 2), heads[2], num_blocks[2])

        self.up3_2 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level2 = Conv1x1(int(dim * 2 ** 2), int(dim * 2 ** 1), bias=bias)
        self.noise_level2 = TransformerBlock(dim=int(dim * 2 ** 1) + 224, num_heads=heads[2],
                                             ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                             LayerNorm_type=LayerNorm_type)
        self.reduce_noise_level2 = Conv1x1(int(dim * 2 ** 1) + 224, int(dim * 2 ** 2), bias=bias)
        self.decoder_level2 = stage(int(dim * 2 ** 1), heads[1], num_blocks[1])

        self.up2_1 = Upsample(int(dim * 2 ** 1))
        self.noise_level1 = TransformerBlock(dim=int(dim * 2 ** 1) + 64, num_heads=heads[2],
                                             ffn_expansion_factor=ffn_expansion_factor, bias=bias,
                                             LayerNorm_type=LayerNorm_type)
        self.reduce_noise_level1 = Conv1x1(int(dim * 2 ** 1) + 64, int(dim * 2 ** 1), bias=bias)
        self.decoder_level1 = stage(int(dim * 2 ** 1), heads[0], num_blocks[0])
        self.refinement = stage(int(dim * 2 ** 1), heads[0], num_refinement_blocks)
        self.output = Conv3x3(int(dim * 2 ** 1), out_channels, bias=bias)

    def forward(self, inp_img, noise_emb=None):
        if inp_img.dim() != 4:
            raise RuntimeError(f"PromptIR expects a [B, C, H, W] tensor, got {list(inp_img.shape)}")
        if inp_img.shape[2] % 8 or inp_img.shape[3] % 8:
            raise RuntimeError("pixel_unshuffle expects height and width to be divisible by 2 at every level: "
                               f"H, W must be multiples of 8, got {list(inp_img.shape[2:])}")
        ops._require_gpu(inp_img)

        return self.decode(inp_img, *self.encode(inp_img))

    # The forward in three pieces (net/model.py:324-334 | :336-337 | :339-377).  A data-parallel trainer cuts the
    # autograd graph between them so that the gradient all-reduce of a finished piece overlaps the backward of the
    # next one (promptir_amd/train.py); `forward` is their composition.
    def encode_levels(self, inp_img):
        """patch_embed, encoder levels 1-3 and the three downsamplings -> (enc1, enc2, enc3, latent input)."""
        inp_enc_level1 = self.patch_embed(inp_img)
        # The six torch.cat of the decoder (net/model.py:341-370) cost nothing here: each concat buffer is allocated
        # BEFORE its first producer runs and both producers write their halves in place (the last block of an encoder
        # level / the latent / a decoder level through its project_out epilogue, the upsampling through its pixel
        # shuffle, the prompt blocks through their 3x3 convolution); ops.CatChannelsFn then finds the two halves
        # adjacent in a registered buffer and returns the buffer.  A skip tensor is the second half of
        # [upsampled | skip]; latent / decoder outputs are the first half of [features | prompt].
        b, _, h, w = inp_img.shape
        dim = self.patch_embed.proj.out_channels
        place = ops.CAT_INPLACE and self.decoder_level1[0].attn.qkv.bias is None
        # every encoder output feeds the next level AND the decoder's skip connection: ops.fork sums the two
        # gradients with the library's own kernel.  (Each buffer is allocated right before its first producer runs: the
        # registry drops buffers nobody has written into yet at the next allocation.)
        c6 = ops.new_cat_buffer(inp_img, 2 * dim, h, w) if place else None                # [up2_1 | enc1]
        out_enc_level1, skip1 = ops.fork(_run_stage(self.encoder_level1, inp_enc_level1, _slot(c6, dim, dim)))
        x2 = self.down1_2(out_enc_level1)
        c4 = ops.new_cat_buffer(inp_img, 4 * dim, h // 2, w // 2) if place else None      # [up3_2 | enc2]
        out_enc_level2, skip2 = ops.fork(_run_stage(self.encoder_level2, x2, _slot(c4, 2 * dim, 2 * dim)))
        x3 = self.down2_3(out_enc_level2)
        c2 = ops.new_cat_buffer(inp_img, 6 * dim, h // 4, w // 4) if place else None      # [up4_3 | enc3]
        out_enc_level3, skip3 = ops.fork(_run_stage(self.encoder_level3, x3, _slot(c2, 2 * dim, 4 * dim)))
        return skip1, skip2, skip3, self.down3_4(out_enc_level3)

    def run_latent(self, inp_latent):
        if self.decoder and ops.CAT_INPLACE and self.latent[0].attn.qkv.bias is None:
            b, c, h, w = inp_latent.shape
            c1 = ops.new_cat_buffer(inp_latent, c + self.prompt3.conv3x3.out_channels, h, w)   # [latent | prompt3]
            return _run_stage(self.latent, inp_latent, _slot(c1, 0, c))
        return self.latent(inp_latent)

    def encode(self, inp_img):
        out_enc_level1, out_enc_level2, out_enc_level3, inp_latent = self.encode_levels(inp_img)
        return self.run_latent(inp_latent), out_enc_level3, out_enc_level2, out_enc_level1

    def decode(self, inp_img, latent, out_enc_level3, out_enc_level2, out_enc_level1):
        def second_half(first):      # slot behind `first` in the registered buffer it was produced into
            buf = ops.cat_buffer_of(first) if ops.CAT_INPLACE else None
            return _slot(buf, first.shape[1], buf.shape[1] - first.shape[1]) if buf is not None else None

        def first_half(skip):        # slot in front of a skip tensor
            buf = ops.cat_buffer_of(skip) if ops.CAT_INPLACE else None
            return _slot(buf, 0, buf.shape[1] - skip.shape[1]) if buf is not None else None

        def prompt_buffer(x, prompt):   # [decoder features | prompt] allocated before the decoder level runs
            if not (self.decoder and ops.CAT_INPLACE and self.decoder_level1[0].attn.qkv.bias is None):
                return None
            return ops.new_cat_buffer(x, prompt.linear_layer.in_features + prompt.conv3x3.out_channels, x.shape[2], x.shape[3])

        if self.decoder:
            latent, to_prompt = ops.fork(latent)
            latent = _cat(latent, self.prompt3(to_prompt, out=second_half(latent)))
            latent = self.reduce_noise_level3(self.noise_level3(latent))

        inp_dec_level3 = _cat(self.up4_3(latent, out=first_half(out_enc_level3)), out_enc_level3)
        x3 = self.reduce_chan_level3(inp_dec_level3)
        c3 = prompt_buffer(x3, self.prompt2) if self.decoder else None
        out_dec_level3 = _run_stage(self.decoder_level3, x3, _slot(c3, 0, x3.shape[1]))
        if self.decoder:
            out_dec_level3, to_prompt = ops.fork(out_dec_level3)
            out_dec_level3 = _cat(out_dec_level3, self.prompt2(to_prompt, out=second_half(out_dec_level3)))
            out_dec_level3 = self.reduce_noise_level2(self.noise_level2(out_dec_level3))

        inp_dec_level2 = _cat(self.up3_2(out_dec_level3, out=first_half(out_enc_level2)), out_enc_level2)
        x2 = self.reduce_chan_level2(inp_dec_level2)
        c5 = prompt_buffer(x2, self.prompt1) if self.decoder else None
        out_dec_level2 = _run_stage(self.decoder_level2, x2, _slot(c5, 0, x2.shape[1]))
        if self.decoder:
            out_dec_level2, to_prompt = ops.fork(out_dec_level2)
            out_dec_level2 = _cat(out_dec_level2, self.prompt1(to_prompt, out=second_half(out_dec_level2)))
            out_dec_level2 = self.reduce_noise_level1(self.noise_level1(out_dec_level2))

        inp_dec_level1 = _cat(self.up2_1(out_dec_level2, out=first_half(out_enc_level1)), out_enc_level1)
        out_dec_level1 = self.refinement(self.decoder_level1(inp_dec_level1))
        return self.output(out_dec_level1, residual=inp_img)   # `+ inp_img` fused into the conv epilogue

    # which piece owns a parameter (for the trainer's gradient segments): 0 encoder levels, 1 latent, 2 the rest
    @staticmethod
    def stage_of(param_name: str) -> int:
        head = param_name.split(".", 1)[0]
        if head in ("patch_embed", "encoder_level1", "encoder_level2", "encoder_level3", "down1_2", "down2_3", "down3_4"):
            return 0
        if head == "latent":
            return 1
        return 2

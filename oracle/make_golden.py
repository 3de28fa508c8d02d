#!/usr/bin/env python3
"""ORACLE tooling — generates tests/golden/*.npz by running the REAL reference.

Run in the build container only (the reference does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

It imports /root/reference/net/model.py (and utils/schedulers.py) read-only,
loads this repo's deterministic weights (promptir_amd/weights.py) into the
reference modules and stores inputs, outputs and gradient summaries.  Only data
is written; no reference source or bytecode is copied.

For large gradient tensors two scalars are stored instead of the tensor:
its L2 norm and its dot product with a fixed probe vector
(uniform01(name + "#probe") - 0.5), both accumulated in float64.
"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from promptir_amd import weights as W  # noqa: E402


def _reference(relpath: str):
    """Load a module of the REAL reference by file path.  (`import net.model` would resolve to this repo's own
    drop-in `net/` package: a regular package shadows the reference's namespace package whatever sys.path says.)"""
    import importlib.util

    name = "reference_" + relpath.replace("/", "_").removesuffix(".py")
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    assert os.path.realpath(mod.__file__).startswith(REF + "/"), mod.__file__
    return mod

OUT = os.path.join(REPO, "tests", "golden")
FULL_GRAD_MAX = 4096


def load_generated(module: torch.nn.Module, seed: int, prefix: str = "") -> None:
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = {k: torch.from_numpy(W.make_tensor(prefix + k, s, seed)) for k, s in shapes.items()}
    module.load_state_dict(sd, strict=True)


def probe(name: str, count: int) -> np.ndarray:
    return W.uniform01(name + "#probe", count).astype(np.float64) - 0.5


def grad_summary(named_params, out: dict, tag: str) -> None:
    names, norms, dots = [], [], []
    for name, p in named_params:
        names.append(name)
        if p.grad is None:
            norms.append(np.nan)
            dots.append(np.nan)
            continue
        g = p.grad.detach().double().numpy().ravel()
        norms.append(float(np.sqrt((g * g).sum())))
        dots.append(float((g * probe(name, g.size)).sum()))
        if g.size <= FULL_GRAD_MAX:
            out[f"{tag}grad/{name}"] = p.grad.detach().numpy().copy()
    out[f"{tag}grad_names"] = np.array(json.dumps(names))
    out[f"{tag}grad_norm"] = np.array(norms, dtype=np.float64)
    out[f"{tag}grad_probe"] = np.array(dots, dtype=np.float64)


def model_case(fname, ctor_kwargs, batch, height, width, sigma, seed, with_backward=True):
    PromptIR = _reference("net/model.py").PromptIR  # the real reference

    torch.manual_seed(0)
    net = PromptIR(**ctor_kwargs)
    load_generated(net, seed)
    degraded, clean = W.synthetic_pair(batch, height, width, sigma=sigma, seed=seed)
    x = torch.from_numpy(degraded)
    t = torch.from_numpy(clean)
    out = {"x": degraded, "clean": clean,
           "ctor": np.array(json.dumps(ctor_kwargs)), "seed": np.array(seed)}
    if with_backward:
        y = net(x)
        loss = torch.nn.L1Loss()(y, t)
        loss.backward()
        out["loss"] = np.array(float(loss.detach()), dtype=np.float64)
        grad_summary(list(net.named_parameters()), out, "")
    else:
        with torch.no_grad():
            y = net(x)
    out["y"] = y.detach().numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "y", tuple(y.shape), "absmax", float(y.abs().max()))


def model_case_chunked(fname, ctor_kwargs, batch, chunk, height, width, sigmas, seed, weights_seed=None):
    """Loss + gradient summaries of a batch too large to run through the reference in one piece in this
    container (64 GiB, no swap): no op of the network mixes samples and nn.L1Loss is a mean over all elements, so
    loss_B = sum_c (n_c / B) loss_c and the gradients add the same way.  The real reference runs on `chunk`
    samples at a time, each chunk's loss scaled by n_c / B before backward(); autograd accumulates in .grad.
    Inputs are not stored: tests rebuild them with W.synthetic_pair(batch, ..., sigma=sigmas, seed=seed)."""
    PromptIR = _reference("net/model.py").PromptIR  # the real reference

    net = PromptIR(**ctor_kwargs)
    load_generated(net, seed if weights_seed is None else weights_seed)
    degraded, clean = W.synthetic_pair(batch, height, width, sigma=sigmas, seed=seed)
    total = 0.0
    ysum = []
    for lo in range(0, batch, chunk):
        x = torch.from_numpy(degraded[lo:lo + chunk])
        t = torch.from_numpy(clean[lo:lo + chunk])
        y = net(x)
        loss = torch.nn.L1Loss()(y, t) * (x.shape[0] / batch)
        loss.backward()
        total += float(loss.detach().double())
        ysum += [float(v) for v in y.detach().double().sum(dim=(1, 2, 3))]
        del y, loss
    out = {"ctor": np.array(json.dumps(ctor_kwargs)), "seed": np.array(seed), "batch": np.array(batch),
           "weights_seed": np.array(seed if weights_seed is None else weights_seed),
           "size": np.array([height, width]), "sigmas": np.array(sigmas, dtype=np.int64),
           "loss": np.array(total, dtype=np.float64), "y_sum": np.array(ysum, dtype=np.float64)}
    grad_summary(list(net.named_parameters()), out, "")
    for k in [k for k in out if k.startswith("grad/")]:   # summaries only: keep the fixture small
        if out[k].size > 256:
            del out[k]
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "loss", total)


def bench_loss_case():
    """Step-1 L1 loss of bench.py's synthetic batches through the REAL reference (forward only, 8 patches at a
    time): weights seed 0; inputs W.synthetic_pair(B, 128, 128, sigma mix, seed=100 + rank).  bench.py asserts
    its first training step reproduces these (config 3: batch 32, rank 0; config 5: batch 8, ranks 0..7)."""
    PromptIR = _reference("net/model.py").PromptIR

    net = PromptIR(decoder=True)
    load_generated(net, 0)
    out = {}
    with torch.no_grad():
        for batch, ranks in ((32, (0,)), (8, tuple(range(8)))):
            for rank in ranks:
                sigmas = [(15, 25, 50)[i % 3] for i in range(batch)]
                degraded, clean = W.synthetic_pair(batch, 128, 128, sigma=sigmas, seed=100 + rank)
                tot = 0.0
                for lo in range(0, batch, 8):
                    y = net(torch.from_numpy(degraded[lo:lo + 8]))
                    tot += float((y - torch.from_numpy(clean[lo:lo + 8])).abs().double().sum())
                out[f"b{batch}_rank{rank}"] = tot / (batch * 3 * 128 * 128)
                print("bench loss", batch, rank, out[f"b{batch}_rank{rank}"])
    with open(os.path.join(OUT, "bench_step1_loss.json"), "w") as f:
        json.dump(out, f, indent=1)


def module_case(store, tag, module, x_shape, seed, call=None):
    load_generated(module, seed, prefix=tag + "/")
    n = int(np.prod(x_shape))
    x = torch.from_numpy((W.uniform01(tag + "#x", n, seed).reshape(x_shape) * 2 - 1).astype(np.float32))
    x.requires_grad_(True)
    y = module(x) if call is None else call(module, x)
    dy = torch.from_numpy((W.uniform01(tag + "#dy", y.numel(), seed).reshape(tuple(y.shape)) * 2 - 1)
                          .astype(np.float32))
    y.backward(dy)
    # x and dy are regenerated by the tests from the same streams (see module_inputs)
    store[f"{tag}/x_shape"] = np.array(x_shape)
    store[f"{tag}/param_shapes"] = np.array(json.dumps({k: list(v.shape) for k, v in module.state_dict().items()}))
    store[f"{tag}/y"] = y.detach().numpy()
    store[f"{tag}/dx"] = x.grad.numpy()
    grad_summary(list(module.named_parameters()), store, tag + "/")
    print("module", tag, tuple(x_shape), "->", tuple(y.shape))


def modules():
    R = _reference("net/model.py")

    store = {}
    seed = 7
    for dim, heads, bhw in [(48, 1, (2, 12, 16)), (96, 2, (2, 12, 16)), (192, 4, (1, 8, 8)), (384, 8, (1, 8, 8)),
                            (704, 4, (1, 8, 8)), (320, 4, (1, 8, 8)), (160, 4, (1, 8, 16))]:
        shp = (bhw[0], dim, bhw[1], bhw[2])
        module_case(store, f"attn_{dim}_{heads}", R.Attention(dim, heads, False), shp, seed)
        module_case(store, f"ffn_{dim}", R.FeedForward(dim, 2.66, False), shp, seed)
    module_case(store, "ln_withbias_48", R.LayerNorm(48, "WithBias"), (2, 48, 16, 24), seed)
    module_case(store, "ln_biasfree_48", R.LayerNorm(48, "BiasFree"), (2, 48, 16, 24), seed)
    module_case(store, "tblock_48_1", R.TransformerBlock(48, 1, 2.66, False, "WithBias"), (2, 48, 16, 24), seed)
    module_case(store, "tblock_96_2_biasfree", R.TransformerBlock(96, 2, 2.66, False, "BiasFree"),
                (1, 96, 8, 16), seed)
    module_case(store, "down_48", R.Downsample(48), (2, 48, 16, 24), seed)
    module_case(store, "up_96", R.Upsample(96), (2, 96, 8, 12), seed)
    module_case(store, "patch_embed", R.OverlapPatchEmbed(3, 48), (2, 3, 16, 24), seed)
    # PromptGenBlock: identity resize, 2x down, anisotropic up
    module_case(store, "prompt_64_id", R.PromptGenBlock(16, 5, 16, 24), (2, 24, 16, 16), seed)
    module_case(store, "prompt_64_down", R.PromptGenBlock(16, 5, 16, 24), (2, 24, 8, 8), seed)
    module_case(store, "prompt_aniso", R.PromptGenBlock(20, 5, 8, 12), (2, 12, 9, 11), seed)
    module_case(store, "prompt_up", R.PromptGenBlock(8, 5, 8, 12), (1, 12, 20, 24), seed)
    np.savez_compressed(os.path.join(OUT, "modules.npz"), **store)


FULL_GRADS_AT_DEPTH = (   # large tensors whose FULL gradient is pinned at full depth, one per kind of path
    "latent.7.ffn.project_in.weight",          # 2042 x 384: split-K weight gradient of the 16^2 level
    "noise_level1.attn.qkv.weight",            # 480 x 160 at 64^2: a noise block (4 heads of 40 channels)
    "noise_level3.attn.temperature",           # 4 x 1 x 1 (heads of 176 channels)
    "prompt3.prompt_param",                    # 1 x 5 x 320 x 16 x 16: adjoint of the softmax-weighted prompt mix
    "refinement.3.attn.qkv.weight",            # 288 x 96 at 128^2: LayerNorm-on-load weight gradient
    "decoder_level1.0.attn.project_out.weight",   # 96 x 96 at 128^2: dW_proj through the attn @ v fold
    "encoder_level3.2.ffn.project_out.weight",    # 192 x 510 at 32^2
    "down2_3.body.0.weight",                   # dense 3x3 weight gradient, 48 x 96 x 3 x 3
)


def full_grad_case():
    """The run of model_full_128_bwd.npz again (full depth, 2 x 3 x 128 x 128, seed 6, real reference), keeping the FULL
    gradient of the tensors above - the norm + probe summaries of that file cannot tell two gradients with equal norm
    and equal projection apart."""
    PromptIR = _reference("net/model.py").PromptIR

    seed = 6
    net = PromptIR(decoder=True)
    load_generated(net, seed)
    degraded, clean = W.synthetic_pair(2, 128, 128, sigma=[25, 50], seed=seed)
    loss = torch.nn.L1Loss()(net(torch.from_numpy(degraded)), torch.from_numpy(clean))
    loss.backward()
    ref = np.load(os.path.join(OUT, "model_full_128_bwd.npz"))
    assert abs(float(loss.detach()) - float(ref["loss"])) <= 1e-9, "not the run of model_full_128_bwd.npz"
    grads = dict(net.named_parameters())
    out = {"loss": np.array(float(loss.detach()), dtype=np.float64), "seed": np.array(seed)}
    for name in FULL_GRADS_AT_DEPTH:
        out["grad/" + name] = grads[name].grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "model_full_128_bwd_fullgrads.npz"), **out)
    print("full gradients", {k: v.shape for k, v in out.items() if k.startswith("grad/")})


def modules_r4():
    """dec1 / refinement shape of the MDTA block: C = 96 with ONE head, so c = 96 rows per head - the only
    per-head width above 48 besides the noise blocks.  The 64 x 64 plane puts the block on the attn @ v fold and on the
    LayerNorm-on-load kernels, as at the network's 128^2 level."""
    R = _reference("net/model.py")

    store = {}
    module_case(store, "attn_96_1", R.Attention(96, 1, False), (2, 96, 12, 16), 7)
    module_case(store, "tblock_96_1", R.TransformerBlock(96, 1, 2.66, False, "WithBias"), (1, 96, 64, 64), 7)
    np.savez_compressed(os.path.join(OUT, "modules_r4.npz"), **store)


def tile_case():
    """Reference model under the restated demo.py tile harness (demo.py itself needs
    `lightning`, which is not installed, so the harness is oracle.tile_eval)."""
    PromptIR = _reference("net/model.py").PromptIR
    from oracle.promptir_ref import tile_eval

    kwargs = dict(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    net = PromptIR(**kwargs)
    load_generated(net, 3)
    degraded, clean = W.synthetic_pair(1, 160, 192, sigma=25, seed=3)
    with torch.no_grad():
        y = tile_eval(net, torch.from_numpy(degraded), tile=128, tile_overlap=32)
    np.savez_compressed(os.path.join(OUT, "tile_eval_small_160x192.npz"), x=degraded, clean=clean,
                        y=y.numpy(), ctor=np.array(json.dumps(kwargs)), seed=np.array(3))
    print("tile_eval", tuple(y.shape))


def tile_full_case():
    """BASELINE config 4 at its real size: the REAL reference (full depth) under the restated demo.py harness on a
    1x3x512x512 image, tile 128 / overlap 32 = 25 tiles.  The input is not stored (tests rebuild it with
    W.synthetic_pair(1, 512, 512, sigma=25, seed=9)); only the blended, clamped image is kept."""
    PromptIR = _reference("net/model.py").PromptIR
    from oracle.promptir_ref import tile_eval, tile_starts

    net = PromptIR(decoder=True)
    load_generated(net, 0)
    degraded, clean = W.synthetic_pair(1, 512, 512, sigma=25, seed=9)
    x = torch.from_numpy(degraded)
    with torch.no_grad():
        y = tile_eval(net, x, tile=128, tile_overlap=32)
    assert tile_starts(512, 128, 32) == [0, 96, 192, 288, 384]
    np.savez_compressed(os.path.join(OUT, "tile_eval_full_512.npz"), y=y.numpy(), seed=np.array(9),
                        weights_seed=np.array(0), size=np.array([512, 512]), sigma=np.array(25),
                        ctor=np.array(json.dumps(dict(decoder=True))))
    print("tile_eval full", tuple(y.shape), "absmax", float(y.abs().max()))


def scheduler_case():
    LinearWarmupCosineAnnealingLR = _reference("utils/schedulers.py").LinearWarmupCosineAnnealingLR

    lin = torch.nn.Linear(2, 2)
    opt = torch.optim.AdamW(lin.parameters(), lr=2e-4)
    sch = LinearWarmupCosineAnnealingLR(optimizer=opt, warmup_epochs=15, max_epochs=150)
    lrs = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for epoch in range(0, 151):
            sch.step(epoch)  # train.py:48-50 passes the epoch
            lrs.append(opt.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(OUT, "scheduler_lr.npz"), lr=np.array(lrs, dtype=np.float64))
    print("scheduler", lrs[:3], lrs[15], lrs[-1])


def structure_case():
    """Known-answer structure facts of the default network."""
    PromptIR = _reference("net/model.py").PromptIR

    net = PromptIR(decoder=True)
    sd = net.state_dict()
    shapes = {k: list(v.shape) for k, v in sd.items()}
    with open(os.path.join(OUT, "state_dict_shapes.json"), "w") as f:
        json.dump({"shapes": shapes, "num_params": sum(p.numel() for p in net.parameters())}, f)
    print("structure", len(shapes), "tensors")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = set(sys.argv[1:])

    def want(k):
        return not which or k in which

    if want("structure"):
        structure_case()
    if want("scheduler"):
        scheduler_case()
    if want("modules"):
        modules()
    small = dict(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    if want("small"):
        model_case("model_small_64.npz", small, 2, 64, 64, [15, 50], 1)
        model_case("model_small_72x88.npz", small, 1, 72, 88, 25, 2)
        model_case("model_small_128.npz", small, 1, 128, 128, 25, 4, with_backward=False)
    if want("tile"):
        tile_case()
    if want("tilefull"):   # BASELINE config 4 at full size and depth (25 tiles through the real reference)
        tile_full_case()
    if want("config3b32"):   # BASELINE config 3 at its real batch: bench.py's own batch (seed 100, rank 0), chunks of 4
        model_case_chunked("model_full_128_b32.npz", dict(decoder=True), 32, 4, 128, 128,
                           [(15, 25, 50)[i % 3] for i in range(32)], 100, weights_seed=0)
    if want("full"):
        model_case("model_full_64.npz", dict(decoder=True), 1, 64, 64, 25, 0)
        model_case("model_full_128.npz", dict(decoder=True), 1, 128, 128, 25, 5, with_backward=False)
    if want("bias"):   # bias=True (net/model.py:253): never used by the reference's callers, part of the ctor surface
        model_case("model_small_64_bias.npz", dict(small, bias=True), 1, 64, 64, 25, 12)
        R = _reference("net/model.py")
        store = {}
        module_case(store, "attn_48_1_bias", R.Attention(48, 1, True), (2, 48, 12, 16), 7)
        module_case(store, "ffn_48_bias", R.FeedForward(48, 2.66, True), (2, 48, 12, 16), 7)
        module_case(store, "tblock_96_2_bias", R.TransformerBlock(96, 2, 2.66, True, "WithBias"), (1, 96, 8, 16), 7)
        np.savez_compressed(os.path.join(OUT, "modules_bias.npz"), **store)
        net = R.PromptIR(decoder=True, bias=True)
        with open(os.path.join(OUT, "state_dict_shapes_bias.json"), "w") as f:
            json.dump({"keys": list(net.state_dict().keys()),
                       "num_params": sum(p.numel() for p in net.parameters())}, f)
    if want("benchloss"):
        bench_loss_case()
    if want("modules_r4"):
        modules_r4()
    if want("fullgrads"):
        full_grad_case()
    if want("config3"):   # BASELINE config 3 shapes (128x128, full depth) WITH backward
        model_case("model_full_128_bwd.npz", dict(decoder=True), 2, 128, 128, [25, 50], 6)
        model_case_chunked("model_full_128_b16.npz", dict(decoder=True), 16, 4, 128, 128,
                           [(15, 25, 50)[i % 3] for i in range(16)], 8)


if __name__ == "__main__":
    main()

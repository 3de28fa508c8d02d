"""The C-ABI library loads here (no GPU) and exports every symbol include/promptir_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "promptir_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pir_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from promptir_amd import _lib

    names = _declared()
    assert len(names) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in the header but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert _lib.lib.pir_arch() == b"gfx950"
    assert _lib.lib.pir_abi_version() == _lib.ABI_VERSION
    # the LOADED library describes its own build (compile-time state, not a constant): no diagnostic macro, the ABI
    # version of the binding, compiled for gfx950 - a stale or foreign .so behind PIR_LIB fails here
    flags = _lib.lib.pir_build_flags()
    assert flags & 1 == 0, "diagnostic build"
    assert (flags >> 8) & 0xFFFF == _lib.ABI_VERSION
    assert flags & (1 << 24), "not built with --offload-arch=gfx950"
    assert _lib.lib.pir_tune_set(15, 1) == -22          # round 2's "skip the reductions" experiment knob is gone
    for src in ("gemm.hip", "gemm_x3.hip", "gemm_common.h"):
        text = open(os.path.join(ROOT, "promptir_amd", "csrc", src)).read()
        assert "ABLATE" not in text and "X3_TRACE" not in text, src


def test_host_side_argument_checks_need_no_gpu():
    from promptir_amd import _lib

    # NULL pointers / bad sizes are rejected on the host before any launch
    assert _lib.lib.pir_add(None, None, None, 10, None) == -22
    assert _lib.lib.pir_gemm_nt_ws_floats(0, 4, 4, 1, 1) == 0
    assert _lib.lib.pir_gemm_nt_ws_floats(48, 48, 16384, 8, 1) > 0
    g = _lib.GemmNN()
    assert _lib.lib.pir_gemm_nn(ctypes.byref(g), None) == -22
    assert _lib.lib.pir_bias_add(None, 0, None, 1, 1, 1, None) == -22


def test_fastdiv_bounds_and_conv3x3_guard():
    """pir_fastdiv(n, magic(d)) = umulhi(n, floor(2^32/d) + 1) is exact only while n * d < 2^32 (pir_common.h).
    The bf16x3 dense-3x3 kernel divides pixel indices by the image width: images where that bound does not hold
    (W=2048, H>1024, first wrong quotient at n=2099199) must be refused, not mis-computed."""
    import numpy as np

    from promptir_amd import _lib

    def magic(d):
        return 0 if d <= 1 else (1 << 32) // d + 1

    def fastdiv(n, m):
        return (n.astype(np.uint64) * np.uint64(m)) >> np.uint64(32) if m else n

    for d in (3, 7, 48, 127, 128, 255, 510, 1021, 2048):
        top = (1 << 32) // d
        n = np.unique(np.concatenate([np.arange(0, min(top, 1 << 16)), np.arange(max(top - 4096, 0), top),
                                      (np.arange(1, 4097) * d - 1).clip(0, top - 1), np.arange(1, 4097) * d % top]))
        assert np.array_equal(fastdiv(n, magic(d)), n // d), d
    n = np.array([2099199], dtype=np.int64)          # row 1024, column 2047 of a 2048-wide image
    assert int(fastdiv(n, magic(2048))[0]) != 2099199 // 2048
    # host-side guard: rejected before any launch (dummy non-null pointers are never dereferenced)
    fake = ctypes.c_void_p(256)
    assert _lib.lib.pir_conv3x3_x3(fake, 16, fake, 3 * 2048 * 1100, fake, 48 * 2048 * 1100, None, 0,
                                   1, 48, 3, 1100, 2048, None) == -22


def test_module_has_no_cpu_fallback():
    import pytest
    import torch

    from net.model import PromptIR

    net = PromptIR(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.zeros(1, 3, 64, 64))


def test_state_dict_matches_reference_layout():
    import json

    from net.model import PromptIR

    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes.json")))
    net = PromptIR(decoder=True)
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref["shapes"].keys())
    assert all(list(sd[k].shape) == ref["shapes"][k] for k in sd)
    assert sum(p.numel() for p in net.parameters()) == ref["num_params"] == 35_592_263
    # bias=True (net/model.py:253): same registration order as nn.Conv2d (weight, bias)
    refb = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes_bias.json")))
    netb = PromptIR(decoder=True, bias=True)
    assert list(netb.state_dict().keys()) == refb["keys"]
    assert sum(p.numel() for p in netb.parameters()) == refb["num_params"]


def test_checkpoint_interchange_with_the_lightning_layout():
    """a reference / Lightning checkpoint stores a torch.optim.AdamW state_dict under
    `optimizer_states[0]`; resuming from one must work, and the checkpoints written here must load into a real
    torch.optim.AdamW over the module's parameters (what Lightning does on `fit(ckpt_path=...)`)."""
    import torch

    from net.model import PromptIR
    from promptir_amd.train import DataParallelTrainer, FlatAdamW, load_lightning_checkpoint

    kw = dict(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    torch.manual_seed(0)
    ref_net = PromptIR(**kw)
    # a reference-layout checkpoint: AdamW over ALL parameters, state only for those that received gradients
    params = list(ref_net.parameters())
    names = [n for n, _ in ref_net.named_parameters()]
    opt = torch.optim.AdamW(params, lr=2e-4)
    for n, p in zip(names, params):
        if not n.startswith(("chnl_reduce", "reduce_noise_channel_")):
            p.grad = torch.randn_like(p) * 1e-3
    opt.step(); opt.step()
    ckpt = {"epoch": 3, "global_step": 2, "pytorch-lightning_version": "2.0.1",
            "state_dict": {"net." + k: v.clone() for k, v in ref_net.state_dict().items()},
            "optimizer_states": [opt.state_dict()], "lr_schedulers": [{}], "loops": {}, "callbacks": {}}

    net = PromptIR(**kw)
    load_lightning_checkpoint(net, ckpt)
    flat = FlatAdamW(net)
    flat.load_state_dict(ckpt["optimizer_states"][0], net)
    assert flat.steps == 2
    sd = opt.state_dict()["state"]
    for i, n in enumerate(names):
        if i in sd:
            o, k = flat.offsets[n], params[i].numel()
            assert torch.equal(flat.exp_avg[o:o + k], sd[i]["exp_avg"].reshape(-1)), n
            assert torch.equal(flat.exp_avg_sq[o:o + k], sd[i]["exp_avg_sq"].reshape(-1)), n
        else:
            assert n not in flat.offsets
    # and back: what this repo writes loads into torch's AdamW
    trainer = DataParallelTrainer.__new__(DataParallelTrainer)
    trainer.net, trainer.opt = net, flat
    out = trainer.checkpoint(epoch=3)
    assert out["pytorch-lightning_version"] == "2.0.1" and "loops" in out and "callbacks" in out and "lr_schedulers" in out
    assert list(out["state_dict"].keys()) == ["net." + k for k in ref_net.state_dict().keys()]
    opt2 = torch.optim.AdamW(list(PromptIR(**kw).parameters()), lr=2e-4)
    opt2.load_state_dict(out["optimizer_states"][0])
    st2 = opt2.state_dict()["state"]
    assert set(st2.keys()) == set(sd.keys())
    for i in sd:
        assert torch.equal(st2[i]["exp_avg"], sd[i]["exp_avg"]) and float(st2[i]["step"]) == 2.0
    # Scheduler block as Lightning 2.0.1 stores it at the epoch-3 save.  Source order restated (Lightning is
    # not installed here): `_TrainingEpochLoop.advance` ends with
    #     if self._num_ready_batches_reached():
    #         self.update_lr_schedulers("epoch", update_plateau_schedulers=False)
    # and only afterwards `FitLoop.on_advance_end` calls the `on_train_epoch_end` hooks (ModelCheckpoint) and then
    # `update_lr_schedulers("epoch", update_plateau_schedulers=True)` (plateau schedulers only).  The reference's
    # `lr_scheduler_step` passes `self.current_epoch` (= 3 here), so the saved scheduler has last_epoch 3, has been
    # stepped 1 (construction) + 4 times, and `_last_lr` / the optimizer lr are closed_form(3), the rate of epoch 4.
    from promptir_amd.train import lightning_epoch_lr, warmup_cosine_lr
    lr3 = warmup_cosine_lr(3)
    assert opt2.state_dict()["param_groups"][0]["lr"] == lr3
    sch = out["lr_schedulers"][0]
    assert sch["last_epoch"] == 3 and sch["_step_count"] == 5 and sch["_last_lr"] == [lr3]
    assert sch["_step_count"] - 2 == sch["last_epoch"]
    assert lightning_epoch_lr(4) == lr3 and lightning_epoch_lr(0) == 0.0 and lightning_epoch_lr(1) == 0.0
    e0 = trainer.checkpoint(epoch=0)["lr_schedulers"][0]
    assert e0["last_epoch"] == 0 and e0["_step_count"] == 2 and e0["_last_lr"] == [0.0]
    assert out["optimizer_states"][0]["param_groups"][0]["initial_lr"] == 2e-4
    # demo.py / evaluate.py read checkpoints through load_checkpoint_file (non-tensor payload, torch >= 2.6)
    for drv in ("demo.py", "evaluate.py"):
        text = open(os.path.join(ROOT, drv)).read()
        assert "load_checkpoint_file(ckpt_path)" in text and "torch.load(" not in text, drv


def test_flat_optimizer_state_written_in_another_order_is_remapped_by_name():
    """round-1 checkpoints hold the flat AdamW moments in named_parameters() order; this engine lays
    the flat buffers out stage by stage.  Same total size, different positions: loading must remap by name (and refuse
    states that do not cover the module), never copy verbatim."""
    import pytest
    import torch

    from net.model import PromptIR
    from promptir_amd.train import FlatAdamW, live_parameters

    kw = dict(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    net = PromptIR(**kw)
    flat = FlatAdamW(net)
    # the round-1 layout: parameters in module order, same 64-float alignment
    old_offsets, off = {}, 0
    for n, p in live_parameters(net):
        old_offsets[n] = off
        off += (p.numel() + FlatAdamW.ALIGN - 1) // FlatAdamW.ALIGN * FlatAdamW.ALIGN
    assert off == flat.numel and old_offsets != flat.offsets          # the case the advisor describes
    g = torch.Generator().manual_seed(3)
    old = {"steps": 7, "exp_avg": torch.randn(off, generator=g), "exp_avg_sq": torch.rand(off, generator=g),
           "offsets": old_offsets}
    flat.load_state_dict(old)
    assert flat.steps == 7
    for n, p in flat.named:
        k, o, src = p.numel(), flat.offsets[n], old_offsets[n]
        assert torch.equal(flat.exp_avg[o:o + k], old["exp_avg"][src:src + k]), n
        assert torch.equal(flat.exp_avg_sq[o:o + k], old["exp_avg_sq"][src:src + k]), n
    # own layout round-trips verbatim
    again = FlatAdamW(PromptIR(**kw))
    again.load_state_dict(flat.state_dict())
    assert torch.equal(again.exp_avg, flat.exp_avg) and again.steps == 7
    bad = dict(old, offsets={k: v for k, v in list(old_offsets.items())[1:]})
    with pytest.raises(ValueError):
        flat.load_state_dict(bad)



def test_gemm_plan_reaches_the_tuned_tiles_for_config3_shapes():
    """BASELINE config 3 (128x128 patches): the GDFN project_in pair of the 96-channel levels (dec1 / refinement) runs
    on tiles tuned for it (gemm_x3.hip, pir_nn_x3_plan: 128 x 128 forward, 96 x 128 input gradient).  Pin that those
    exact shapes select them; the GPU parity of the same shapes is
    tests/test_kernels_gpu.py::test_conv1x1_config3_shapes."""
    from promptir_amd import _lib

    def plan(M, K, N, batch, presplit=True):
        g = _lib.GemmNN()
        g.M, g.K, g.N, g.O1, g.O2, g.ldx, g.ldy = M, K, N, batch, 1, N, N
        g.A3 = 256 if presplit else None
        g.a3_kp = (K + 15) // 16 * 16 if presplit else 0
        return _lib.lib.pir_gemm_nn_plan(ctypes.byref(g))

    # round 3: long pixel streams against short k run on the persistent kernels of gemm_res.hip
    assert plan(510, 96, 16384, 32) == 9100      # project_in forward, dec1 / refinement: B-stationary (activations split once)
    assert plan(510, 96, 4096, 32) == 9100       # level 2 forward
    assert plan(254, 48, 16384, 32) == 9100      # level 1 encoder
    assert plan(255, 96, 4096, 32) == 9100       # project_out input gradient, level 2
    assert plan(288, 96, 16384, 32) == 9000      # qkv forward, dec1 / refinement: resident weight panel, three row tiles
    assert plan(510, 96, 16384, 2) == 2222       # a test-sized batch has too few column blocks per workgroup: 128 x 128 tiles
    assert plan(96, 510, 16384, 32) == 9200      # project_in input gradient (M = 96, K = 510): C-stationary (gemm_cst.hip)
    assert plan(96, 255, 4096, 32) == 9200       # project_out forward of the GDFN, level 2
    assert plan(192, 1020, 1024, 32) == 3114     # 192 rows: C-stationary only inside the fused LayerNorm backward
    assert plan(255, 96, 16384, 32) == 2222      # project_out input gradient at 128^2 (A/B: no gain from the persistent kernels)
    assert plan(96, 510, 16384, 2) == 9200       # the test-sized batch takes the same branch
    assert plan(96, 96, 16384, 32) == 3214       # short k stays on the tiled kernel (96 x 256)
    assert plan(48, 48, 16384, 32) == 1222
    assert plan(48, 144, 16384, 32) == 9200      # 48 rows against k >= 128: C-stationary, 64-row tiles with zero padding rows
    assert plan(48, 254, 16384, 32) == 9200
    assert plan(576, 192, 1024, 32) == 3114      # low-resolution levels: 96 x 128 ...
    assert plan(384, 2042, 256, 32) == 3114
    assert plan(1020, 192, 1024, 32) == 2222     # ... unless 128-row tiles pad less
    assert plan(2042, 384, 256, 32) == 2222
    assert plan(288, 96, 4096, 32) == 3114       # level-2 qkv
    assert plan(96, 510, 16384, 32, presplit=False) != 3114   # the branch needs pre-split weights
    assert plan(0, 1, 1, 1) == -22
    # operands beyond the 32-bit byte offsets of the bf16x3 kernel fall back to the fp32 kernel
    assert plan(96, 255, 2200 * 2048, 1) == 0


def test_part_batch_rule_and_new_entry_points_are_declared():
    """Host logic of round 4 that needs no GPU: how a batch is cut into part streams, and that the header, the binding
    and the library agree on the entry points added this round."""
    from promptir_amd import _lib
    from promptir_amd.train import DataParallelTrainer

    tr = DataParallelTrainer.__new__(DataParallelTrainer)
    tr.micro_streams, tr.min_part = 2, 0
    assert [tr._nparts(b) for b in (1, 4, 7, 8, 16, 32)] == [1, 1, 1, 2, 2, 2]
    tr.micro_streams = 4
    assert [tr._nparts(b) for b in (4, 8, 16, 24, 32, 64)] == [1, 2, 2, 3, 4, 4]
    tr.min_part = 2                                   # PIR_MIN_PART override
    assert tr._nparts(8) == 4
    names = set(_declared())
    for fn in ("pir_reduce_defer", "pir_reduce_flush", "pir_reduce_pending", "pir_reduce_defer_limit", "pir_gemm_nt_group",
               "pir_gemm_nt_ws_needed", "pir_gemm_nt_group_ws_needed", "pir_copy_strided4", "pir_crop_augment_u8",
               "pir_gdfn_fused_fwd", "pir_gdfn_fused_ws_bytes"):
        assert fn in names and fn in _lib.SIGNATURES, fn
    # host-only queries answer without a device
    assert _lib.lib.pir_reduce_defer_limit(-1) == 4 << 20
    assert _lib.lib.pir_reduce_pending(None) == 0
    assert _lib.lib.pir_gdfn_fused_ws_bytes(2, 96, 128, 128) == 2 * 128 * 128 * 96 * 6 + 16384
    assert _lib.lib.pir_gemm_nt_ws_needed(None) == 0

"""All-in-one data readers: the reference's folder layout, sample list, name rules, crop and
augmentation semantics on a tiny generated folder.  CPU-only (PIL + numpy); the device-side kernel is checked against the
same host statement in tests/test_data_gpu.py."""
import os

import numpy as np
import pytest

from promptir_amd import data as D

PIL = pytest.importorskip("PIL.Image")


def _img(path, h, w, seed):
    rng = np.random.RandomState(seed)
    a = rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    PIL.fromarray(a).save(path)
    return a


def make_tree(root):
    """data_dir/{noisy/denoise.txt, rainy/rainTrain.txt, hazy/hazy_outside.txt} + Train/{Denoise, Derain/{rainy,gt},
    Dehaze/{synthetic,original}} as options.py:20-27 / INSTALL.md lay them out."""
    r = str(root) + "/"
    imgs = {}
    for k, name in enumerate(["a.png", "b.png", "c.png"]):          # c.png is in the folder but not in denoise.txt
        imgs[name] = _img(r + "Train/Denoise/" + name, 70 + 3 * k, 85 + k, k)
    os.makedirs(r + "data_dir/noisy", exist_ok=True)
    open(r + "data_dir/noisy/denoise.txt", "w").write("a.png\nb.png\n")
    os.makedirs(r + "data_dir/rainy", exist_ok=True)
    open(r + "data_dir/rainy/rainTrain.txt", "w").write("rainy/rain-1.png\nrainy/rain-2.png\n")
    for n in (1, 2):
        imgs[f"rain-{n}"] = _img(r + f"Train/Derain/rainy/rain-{n}.png", 66, 81, 10 + n)
        imgs[f"norain-{n}"] = _img(r + f"Train/Derain/gt/norain-{n}.png", 66, 81, 20 + n)
    os.makedirs(r + "data_dir/hazy", exist_ok=True)
    open(r + "data_dir/hazy/hazy_outside.txt", "w").write("synthetic/0042_0.8_0.2.png\n")
    imgs["hazy"] = _img(r + "Train/Dehaze/synthetic/0042_0.8_0.2.png", 64, 96, 31)
    imgs["nonhazy"] = _img(r + "Train/Dehaze/original/0042.png", 64, 96, 32)
    return r, imgs


def test_name_rules_and_index_map():
    assert D.rainy_gt_name("/d/Derain/rainy/rain-17.png") == "/d/Derain/gt/norain-17.png"        # dataset_utils.py:113-115
    assert D.nonhazy_name("/d/Dehaze/synthetic/0042_0.8_0.2.jpg") == "/d/Dehaze/original/0042.jpg"   # :117-122
    img = np.arange(7 * 9 * 3, dtype=np.uint8).reshape(7, 9, 3)
    assert D.crop_img(np.zeros((70, 85, 3)), 16).shape == (64, 80, 3)
    assert np.array_equal(D.crop_img(img, 4), img[1:5, 0:8])                                       # image_utils.py:59-64
    # the device kernel's index map == numpy flipud / rot90 of utils/image_utils.py:133-160, all 8 modes
    P = 6
    patch = np.arange(P * P * 3, dtype=np.int32).reshape(P, P, 3)
    for mode in range(8):
        want = D.augment(patch, mode)
        got = np.empty_like(patch)
        for i in range(P):
            for j in range(P):
                si, sj = D.aug_source_index(mode, i, j, P)
                got[i, j] = patch[si, sj]
        assert np.array_equal(got, want), mode


def test_prompt_train_set_over_the_reference_layout(tmp_path):
    r, imgs = make_tree(tmp_path)
    ds = D.PromptTrainSet(r + "data_dir/", r + "Train/Denoise/", r + "Train/Derain/", r + "Train/Dehaze/", patch_size=32, seed=3)
    # sample list as _init_*_ids / _merge_ids build it: 2 clean x 3 repeats x 3 sigmas, 2 rainy x 120, 1 hazy
    assert len(ds) == 2 * 3 * 3 + 2 * 120 + 1
    kinds = [s["de_type"] for s in ds.sample_ids]
    assert kinds[:18] == [0] * 6 + [1] * 6 + [2] * 6 and kinds[18:258] == [3] * 240 and kinds[258:] == [4]
    assert all("c.png" not in s["clean_id"] for s in ds.sample_ids)
    it = ds[0]
    assert it["de_id"] == 0 and it["degraded"] is None and it["name"] in ("a", "b")
    assert it["clean"].shape[0] % 16 == 0 and it["clean"].shape[1] % 16 == 0 and 1 <= it["mode"] <= 7
    assert ds[0]["top"] == it["top"] and ds[0]["mode"] == it["mode"]            # per-item stream: reproducible
    rain = ds[18]
    assert rain["de_id"] == 3 and rain["name"].endswith("gt/norain-1.png")
    assert np.array_equal(rain["clean"], D.crop_img(imgs["norain-1"], 16)) and np.array_equal(rain["degraded"], D.crop_img(imgs["rain-1"], 16))
    haze = ds[258]
    assert haze["de_id"] == 4 and np.array_equal(haze["clean"], D.crop_img(imgs["nonhazy"], 16))
    # host statement of one item: crop + augmentation + ToTensor; paired samples share window and mode
    deg, clean = D.crop_augment_host(rain, 32)
    t, l, m = rain["top"], rain["left"], rain["mode"]
    assert np.array_equal(clean, D.augment(rain["clean"][t:t + 32, l:l + 32], m).transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    assert np.array_equal(deg, D.augment(rain["degraded"][t:t + 32, l:l + 32], m).transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    dn, cl = D.crop_augment_host(it, 32)
    assert dn.shape == cl.shape == (3, 32, 32) and 0.03 < float(np.abs(dn - cl).mean()) < 0.08       # sigma 15 / 255
    # ragged batch: aligned offsets, paired slots, the table the kernel reads
    batch = D.ragged_collate([it, rain, haze])
    meta = batch["meta"].numpy()
    assert meta.shape == (3, 8) and meta[0, 1] == -1 and meta[1, 1] > meta[1, 0] and all(meta[:, 0] % 256 == 0)
    buf = batch["images"].numpy()
    h, w = rain["clean"].shape[:2]
    assert np.array_equal(buf[meta[1, 0]:meta[1, 0] + h * w * 3].reshape(h, w, 3), rain["clean"])
    assert np.array_equal(buf[meta[1, 1]:meta[1, 1] + h * w * 3].reshape(h, w, 3), rain["degraded"])
    assert list(meta[1, 2:7]) == [h, w, t, l, m] and batch["de_id"].tolist() == [0, 3, 4]
    # a DataLoader with workers yields the same batches as direct indexing
    import torch

    loader = torch.utils.data.DataLoader(ds, batch_size=3, sampler=[0, 18, 258], num_workers=2, collate_fn=D.ragged_collate)
    got = next(iter(loader))
    assert torch.equal(got["meta"], batch["meta"]) and torch.equal(got["images"], batch["images"])
    # denoise-only training (de_type subset)
    assert len(D.PromptTrainSet(r + "data_dir/", r + "Train/Denoise/", "", "", ["denoise_25"], 32)) == 6


def test_derain_dehaze_test_set(tmp_path):
    r = str(tmp_path) + "/"
    a = _img(r + "derain/Rain100L/input/rain-001.png", 50, 70, 1)
    b = _img(r + "derain/Rain100L/target/rain-001.png", 50, 70, 2)
    c = _img(r + "dehaze/input/0007_0.9_0.16.png", 48, 64, 3)
    d = _img(r + "dehaze/target/0007.png", 48, 64, 4)
    ds = D.DerainDehazeTestSet(r + "derain/Rain100L/", r + "dehaze/", "derain")
    name, deg, clean = ds[0]
    assert name == "rain-001" and np.array_equal(deg, D.crop_img(a, 16)) and np.array_equal(clean, D.crop_img(b, 16))
    ds.set_dataset("dehaze")
    name, deg, clean = ds[0]
    assert name == "0007_0.9_0.16" and np.array_equal(deg, D.crop_img(c, 16)) and np.array_equal(clean, D.crop_img(d, 16))

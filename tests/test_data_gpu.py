"""Device side of the all-in-one data path: pir_crop_augment_u8 against the host statement (index work: bit-exact)."""
import numpy as np
import pytest
import torch

from promptir_amd import data as D

pytestmark = pytest.mark.gpu


def _items(P):
    rng = np.random.RandomState(5)
    items = []
    for k, mode in enumerate(range(8)):                       # every augmentation mode, ragged image sizes
        h, w = P + 16 * (k % 3), P + 16 * ((k + 1) % 4)
        clean = rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        paired = k % 2 == 1
        items.append({"name": f"s{k}", "de_id": (3 + k % 2) if paired else k % 3, "clean": clean,
                      "degraded": rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8) if paired else None,
                      "top": int(rng.randint(0, h - P + 1)), "left": int(rng.randint(0, w - P + 1)), "mode": mode,
                      "noise_seed": 100 + k})
    return items


@pytest.mark.parametrize("P", [32, 128])
def test_crop_augment_degrade_on_device_equals_the_host_statement(P):
    items = _items(P)
    batch = D.ragged_collate(items)
    dev = torch.device("cuda:0")
    deg, clean = D.crop_augment_gpu(batch["images"].to(dev), batch["meta"], batch["de_id"].tolist(),
                                    batch["noise_seed"].tolist(), P)
    torch.cuda.synchronize()
    for b, it in enumerate(items):
        hd, hc = D.crop_augment_host(it, P)
        assert np.array_equal(clean[b].cpu().numpy(), hc), ("clean", b, it["mode"])       # crop + flip / rot90 + ToTensor
        got = deg[b].cpu().numpy()
        if it["degraded"] is not None:
            assert np.array_equal(got, hd), ("paired", b, it["mode"])
        else:
            # same generator; libm rounding of log / cos may move a value across an integer boundary
            diff = np.abs(got - hd)
            assert float(diff.max()) <= 1.0 / 255 + 1e-7 and float((diff > 0).mean()) <= 1e-3, (b, float(diff.max()))
            sig = D.SIGMA_OF_DE_ID[it["de_id"]]
            assert abs(float((got - hc).std()) * 255 - sig) < 0.35 * sig              # clipped noise of the right scale


def test_ragged_prefetcher_feeds_a_train_step(tmp_path):
    """PromptTrainSet -> DataLoader(ragged_collate) -> RaggedDevicePrefetcher -> trainer: the reference's all-in-one
    input path end to end on a generated folder."""
    from tests.test_data import make_tree

    r, _ = make_tree(tmp_path)
    ds = D.PromptTrainSet(r + "data_dir/", r + "Train/Denoise/", r + "Train/Derain/", r + "Train/Dehaze/", patch_size=32, seed=1)
    idx = [0, 7, 18, 258, 100, 12, 30, 15]
    loader = torch.utils.data.DataLoader(ds, batch_size=4, sampler=idx, num_workers=2, pin_memory=True,
                                         collate_fn=D.ragged_collate, drop_last=True)
    dev = torch.device("cuda:0")
    got = list(D.RaggedDevicePrefetcher(loader, dev, 32))
    assert len(got) == 2
    for k, (deg, clean) in enumerate(got):
        assert deg.shape == clean.shape == (4, 3, 32, 32)
        for b in range(4):
            hd, hc = D.crop_augment_host(ds[idx[4 * k + b]], 32)
            assert np.array_equal(clean[b].cpu().numpy(), hc)
            if ds[idx[4 * k + b]]["degraded"] is not None:
                assert np.array_equal(deg[b].cpu().numpy(), hd)

"""Training / evaluation data for the PromptIR path.

The reference's datasets (BSD400+WED, Rain100L, RESIDE, BSD68 ...) are not available offline, so the
default source is a deterministic synthetic one with the same item structure as
`PromptTrainDataset.__getitem__` (utils/dataset_utils.py:133-172): ([clean_name, de_id], degrad_patch,
clean_patch) with de_id 0/1/2 = Gaussian noise sigma 15/25/50 added in the uint8 domain
(utils/degradation_utils.py:21-37).  `degrade_gaussian_gpu` performs that noise injection on the GPU with the
same counter-based generator as `promptir_amd.weights.degrade_gaussian` (host), so both agree.
A directory of real clean images is used instead when `denoise_dir` exists (PIL reader, random crop and the
reference's 7-way flip/rot90 augmentation, utils/image_utils.py:133-182).
"""
from __future__ import annotations

import os
import random
from typing import List, Sequence

import numpy as np
import torch

from . import weights as W

SIGMA_OF_DE_ID = {0: 15.0, 1: 25.0, 2: 50.0}
_GOLDEN = 0x9E3779B97F4A7C15


def _stream_key(name: str, seed: int) -> int:
    return W.fnv1a64(name) ^ ((seed * _GOLDEN) & 0xFFFFFFFFFFFFFFFF)


def degrade_gaussian_gpu(clean: torch.Tensor, sigmas: Sequence[float], seed: int = 0,
                         stream_seeds: Sequence[int] = None) -> torch.Tensor:
    """uint8-domain noise on the device; image i uses the host streams of `weights.synthetic_pair`
    (seed + 1000*(i+1), or `stream_seeds[i]` when given), so results equal weights.degrade_gaussian up to libm
    rounding of log/cos."""
    from .ops import _require_gpu, _stream, check, lib

    _require_gpu(clean)
    clean = clean.contiguous()
    b = clean.shape[0]
    per = clean[0].numel()
    keys = []
    for i in range(b):
        s = seed + 1000 * (i + 1) if stream_seeds is None else int(stream_seeds[i])
        keys += [_stream_key("noise#bm1", s), _stream_key("noise#bm2", s)]
    keys_t = torch.tensor(np.array(keys, dtype=np.uint64).view(np.int64), device=clean.device)
    sig = torch.tensor(list(sigmas), dtype=torch.float32, device=clean.device)
    out = torch.empty_like(clean)
    check(lib.pir_degrade_gaussian(clean.data_ptr(), out.data_ptr(), sig.data_ptr(), keys_t.data_ptr(), per, b, _stream()),
          "pir_degrade_gaussian")
    return out


def augment(patch: np.ndarray, mode: int) -> np.ndarray:
    """utils/image_utils.py:133-160 data_augmentation on HWC arrays, modes 0..7."""
    if mode == 0:
        return patch
    if mode == 1:
        return np.flipud(patch)
    if mode == 2:
        return np.rot90(patch)
    if mode == 3:
        return np.flipud(np.rot90(patch))
    if mode == 4:
        return np.rot90(patch, k=2)
    if mode == 5:
        return np.flipud(np.rot90(patch, k=2))
    if mode == 6:
        return np.rot90(patch, k=3)
    if mode == 7:
        return np.flipud(np.rot90(patch, k=3))
    raise ValueError(mode)


class SyntheticTrainSet(torch.utils.data.Dataset):
    """Deterministic stand-in with PromptTrainDataset's item structure; clean patches are 8-bit valued."""

    def __init__(self, length: int, patch_size: int = 128, de_types: Sequence[int] = (0, 1, 2), seed: int = 0):
        self.length, self.patch, self.de_types, self.seed = length, patch_size, list(de_types), seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        de_id = self.de_types[idx % len(self.de_types)]
        deg, clean = W.synthetic_pair(1, self.patch, self.patch, sigma=[SIGMA_OF_DE_ID[de_id]], seed=self.seed + idx)
        return [f"synthetic_{idx:06d}", de_id], torch.from_numpy(deg[0]), torch.from_numpy(clean[0])


class DenoiseFolderTrainSet(torch.utils.data.Dataset):
    """Real clean images from a folder -> random crop, 7-way augmentation, sigma 15/25/50 noise
    (the de_id < 3 branch of utils/dataset_utils.py:140-156).  Needs PIL; used only when data is supplied."""

    def __init__(self, folder: str, patch_size: int = 128, de_types: Sequence[int] = (0, 1, 2)):
        from PIL import Image  # noqa: F401

        exts = (".png", ".jpg", ".jpeg", ".bmp")
        self.files = sorted(os.path.join(folder, f) for f in os.listdir(folder) if f.lower().endswith(exts))
        if not self.files:
            raise FileNotFoundError(f"no images in {folder}")
        self.patch, self.de_types = patch_size, list(de_types)

    def __len__(self):
        return len(self.files) * len(self.de_types)

    def __getitem__(self, idx):
        from PIL import Image

        path = self.files[idx // len(self.de_types)]
        de_id = self.de_types[idx % len(self.de_types)]
        img = np.array(Image.open(path).convert("RGB"))
        h, w = img.shape[:2]
        img = img[: h - h % 16, : w - w % 16]                       # crop_img(base=16)
        h, w = img.shape[:2]
        top, left = random.randint(0, h - self.patch), random.randint(0, w - self.patch)
        clean = augment(img[top:top + self.patch, left:left + self.patch], random.randint(1, 7)).copy()
        noise = np.random.randn(*clean.shape)
        noisy = np.clip(clean + noise * SIGMA_OF_DE_ID[de_id], 0, 255).astype(np.uint8)
        to_t = lambda a: torch.from_numpy(a.astype(np.float32).transpose(2, 0, 1) / 255.0)
        return [os.path.basename(path).split(".")[0], de_id], to_t(noisy), to_t(clean)


def shard_indices(length: int, rank: int, world: int, epoch: int, seed: int = 0) -> List[int]:
    """DistributedSampler semantics (shuffle with a per-epoch seed, pad to a multiple of world, strided shards)."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    order = torch.randperm(length, generator=g).tolist()
    total = (length + world - 1) // world * world
    order += order[: total - length]
    return order[rank:total:world]


class CleanPatchSet(torch.utils.data.Dataset):
    """Clean 8-bit patches only (what the host side of the GPU degradation path reads): item = (de_id, clean, noise seed).
    The degradation (uint8-domain Gaussian noise, utils/degradation_utils.py:21-27) then runs on the device
    (`degrade_gaussian_gpu`) with the same generator as the host path, so both pipelines produce the same batches."""

    def __init__(self, length: int, patch_size: int = 128, de_types: Sequence[int] = (0, 1, 2), seed: int = 0):
        self.length, self.patch, self.de_types, self.seed = length, patch_size, list(de_types), seed

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        de_id = self.de_types[idx % len(self.de_types)]
        clean = W.synthetic_clean(1, self.patch, self.patch, self.seed + idx)
        clean = np.floor(clean * 255.0).astype(np.float32) / np.float32(255.0)
        return de_id, torch.from_numpy(clean[0]), self.seed + idx


class DevicePrefetcher:
    """Iterates (degraded, clean) device batches one step ahead: the host -> device copies of batch k+1 run on a side
    stream (from pinned memory) while step k computes; replaces DataLoader(pin_memory=True) + Lightning's batch
    transfer (reference train.py:336).  `gpu_degrade`: the loader yields (de_id, clean, seed) and the noise is added
    on the device."""

    def __init__(self, loader, device, gpu_degrade: bool = False):
        self.loader, self.device, self.gpu_degrade = loader, device, gpu_degrade
        self.stream = torch.cuda.Stream(device)

    def _stage(self, item):
        with torch.cuda.stream(self.stream):
            if self.gpu_degrade:
                de_ids, clean, seeds = item
                clean = clean.to(self.device, non_blocking=True)
                sig = [SIGMA_OF_DE_ID[int(d)] for d in de_ids]
                # the generator stream of sample i is the one synthetic_pair(1, ..., seed=s_i) uses: s_i + 1000
                degrad = degrade_gaussian_gpu(clean, sig, stream_seeds=[int(v) + 1000 for v in seeds])
            else:
                _, degrad, clean = item
                degrad = degrad.to(self.device, non_blocking=True)
                clean = clean.to(self.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return degrad, clean, ready

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            degrad, clean, ready = nxt
            try:
                nxt = self._stage(next(it))      # enqueue the next copies before handing out the current batch
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ready)
            degrad.record_stream(torch.cuda.current_stream(self.device))
            clean.record_stream(torch.cuda.current_stream(self.device))
            yield degrad, clean

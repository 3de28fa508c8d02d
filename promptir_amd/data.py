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


# ----------------------------------------------------------------------------------------------------------------
# All-in-one readers: the reference's folder layout on the host, crop + augmentation
# + degradation on the device.
DE_DICT = {'denoise_15': 0, 'denoise_25': 1, 'denoise_50': 2, 'derain': 3, 'dehaze': 4}   # utils/dataset_utils.py:26
IMG_EXTS = ('jpg', 'JPG', 'png', 'PNG', 'jpeg', 'JPEG', 'bmp', 'BMP')


def crop_img(image: np.ndarray, base: int = 64) -> np.ndarray:
    """utils/image_utils.py:59-64: centre crop to multiples of `base`."""
    h, w = image.shape[:2]
    ch, cw = h % base, w % base
    return image[ch // 2:h - ch + ch // 2, cw // 2:w - cw + cw // 2, :]


def rainy_gt_name(rainy_name: str) -> str:
    """utils/dataset_utils.py:113-115: .../rainy/rain-N.png -> .../gt/norain-N.png"""
    return rainy_name.split("rainy")[0] + 'gt/norain-' + rainy_name.split('rain-')[-1]


def nonhazy_name(hazy_name: str) -> str:
    """utils/dataset_utils.py:117-122: .../synthetic/<id>_<a>_<b>.jpg -> .../original/<id>.jpg"""
    dir_name = hazy_name.split("synthetic")[0] + 'original/'
    name = hazy_name.split('/')[-1].split('_')[0]
    suffix = '.' + hazy_name.split('.')[-1]
    return dir_name + name + suffix


def aug_source_index(mode: int, i: int, j: int, P: int):
    """Source (row, column) inside a square P x P patch of output pixel (i, j) of data_augmentation(patch, mode)
    (utils/image_utils.py:133-160): the index map pir_crop_augment_u8 applies; checked against numpy in the tests."""
    q = P - 1
    return {0: (i, j), 1: (q - i, j), 2: (j, q - i), 3: (j, i), 4: (q - i, q - j), 5: (i, q - j), 6: (q - j, i),
            7: (q - j, q - i)}[mode]


class PromptTrainSet(torch.utils.data.Dataset):
    """The reference's PromptTrainDataset (utils/dataset_utils.py:15-175) over the same folder layout and list files:

        <data_file_dir>noisy/denoise.txt      names of the clean images in <denoise_dir>     (x3 per sigma, :47-73)
        <data_file_dir>rainy/rainTrain.txt    paths below <derain_dir> of the rainy images    (x120, :88-97); ground
                                              truth = .../gt/norain-N.* (:113-115)
        <data_file_dir>hazy/hazy_outside.txt  paths below <dehaze_dir> of the hazy images (:77-86); ground truth =
                                              .../original/<id>.* (:117-122)

    Same sample list construction (`sample_ids`, merged in the same order), same crop_img(base=16), same random crop
    window and random_augmentation draw (python `random`: crop :102-111, flag 1..7 utils/image_utils.py:177-182).
    What differs is WHERE the pixels are touched: __getitem__ returns the decoded uint8 image(s) plus the drawn window
    and mode; crop, flip / rot90, ToTensor and the sigma noise run on the device (`crop_augment_gpu`)."""

    def __init__(self, data_file_dir: str, denoise_dir: str, derain_dir: str, dehaze_dir: str,
                 de_type: Sequence[str] = ('denoise_15', 'denoise_25', 'denoise_50', 'derain', 'dehaze'), patch_size: int = 128,
                 seed: int = 0):
        from PIL import Image  # noqa: F401  (needed by __getitem__; fail at construction if absent)

        self.patch, self.seed, self.de_type = patch_size, seed, list(de_type)
        self.sample_ids: List[dict] = []
        rng = random.Random(seed)
        clean_ids = []
        if any(t in self.de_type for t in ('denoise_15', 'denoise_25', 'denoise_50')):
            wanted = {l.strip() for l in open(data_file_dir + "noisy/denoise.txt")}
            clean_ids = [denoise_dir + n for n in sorted(os.listdir(denoise_dir)) if n.strip() in wanted]
            self.num_clean = len(clean_ids)
        for name, de in (('denoise_15', 0), ('denoise_25', 1), ('denoise_50', 2)):
            if name in self.de_type:
                ids = [{"clean_id": x, "de_type": de} for x in clean_ids] * 3
                rng.shuffle(ids)
                self.sample_ids += ids
        if 'derain' in self.de_type:
            rs = [derain_dir + l.strip() for l in open(data_file_dir + "rainy/rainTrain.txt") if l.strip()]
            self.sample_ids += [{"clean_id": x, "de_type": 3} for x in rs] * 120
            self.num_rl = len(rs) * 120
        if 'dehaze' in self.de_type:
            hz = [dehaze_dir + l.strip() for l in open(data_file_dir + "hazy/hazy_outside.txt") if l.strip()]
            self.sample_ids += [{"clean_id": x, "de_type": 4} for x in hz]
            self.num_hazy = len(hz)
        if not self.sample_ids:
            raise FileNotFoundError("no training samples for de_type %s" % (self.de_type,))

    def __len__(self):
        return len(self.sample_ids)

    @staticmethod
    def _read(path: str) -> np.ndarray:
        from PIL import Image

        return crop_img(np.array(Image.open(path).convert('RGB')), base=16)

    def __getitem__(self, idx):
        sample = self.sample_ids[idx]
        de_id = sample["de_type"]
        rng = random.Random((self.seed * 1000003 + idx) & 0xFFFFFFFF)   # per-item stream: worker count does not change the data
        if de_id < 3:
            clean = self._read(sample["clean_id"])
            name = sample["clean_id"].split("/")[-1].split('.')[0]
            degraded = None
        else:
            degraded = self._read(sample["clean_id"])
            name = rainy_gt_name(sample["clean_id"]) if de_id == 3 else nonhazy_name(sample["clean_id"])
            clean = self._read(name)
            if clean.shape != degraded.shape:
                raise ValueError(f"{sample['clean_id']}: degraded {degraded.shape} and ground truth {clean.shape} differ")
        h, w = clean.shape[:2]
        if h < self.patch or w < self.patch:
            raise ValueError(f"{sample['clean_id']}: {h}x{w} is smaller than the {self.patch} patch")
        top, left = rng.randint(0, h - self.patch), rng.randint(0, w - self.patch)
        mode = rng.randint(1, 7)
        noise_seed = self.seed * 7919 + idx
        return {"name": name, "de_id": de_id, "clean": np.ascontiguousarray(clean),
                "degraded": None if degraded is None else np.ascontiguousarray(degraded),
                "top": top, "left": left, "mode": mode, "noise_seed": noise_seed}


def ragged_collate(items):
    """Whole images of different sizes in ONE uint8 buffer (256-byte aligned pieces) + an int64 [B][8] table
    {clean offset, degraded offset or -1, H, W, top, left, mode, 0}: what pir_crop_augment_u8 reads."""
    offs, total = [], 0
    for it in items:
        for key in ("clean", "degraded"):
            a = it[key]
            if a is None:
                offs.append(-1)
                continue
            offs.append(total)
            total += (a.size + 255) // 256 * 256
    buf = torch.empty(max(total, 256), dtype=torch.uint8)
    view = buf.numpy()
    meta = torch.zeros((len(items), 8), dtype=torch.int64)
    for b, it in enumerate(items):
        for k, key in enumerate(("clean", "degraded")):
            a, o = it[key], offs[2 * b + k]
            if a is not None:
                view[o:o + a.size] = a.reshape(-1)
        h, w = it["clean"].shape[:2]
        meta[b] = torch.tensor([offs[2 * b], offs[2 * b + 1], h, w, it["top"], it["left"], it["mode"], 0])
    de = torch.tensor([it["de_id"] for it in items], dtype=torch.int64)
    seeds = torch.tensor([it["noise_seed"] for it in items], dtype=torch.int64)
    return {"images": buf, "meta": meta, "de_id": de, "noise_seed": seeds, "names": [it["name"] for it in items]}


def crop_augment_gpu(images: torch.Tensor, meta: torch.Tensor, de_ids, noise_seeds, patch: int, out=None):
    """(degraded, clean) [B,3,P,P] on the device from a ragged uint8 batch (pir_crop_augment_u8).  `out`: a pair of
    preallocated tensors (e.g. the trainer's static graph inputs) to write into."""
    from .ops import _stream, check, lib

    if not images.is_cuda or images.dtype != torch.uint8 or meta.dtype != torch.int64:
        raise RuntimeError("crop_augment_gpu: uint8 images and int64 meta on a ROCm device expected (no CPU fallback)")
    b = meta.shape[0]
    dev = images.device
    keys, sig = [], []
    for i in range(b):
        s = int(noise_seeds[i])
        keys += [_stream_key("noise#bm1", s), _stream_key("noise#bm2", s)]
        sig.append(SIGMA_OF_DE_ID.get(int(de_ids[i]), 0.0))
    keys_t = torch.tensor(np.array(keys, dtype=np.uint64).view(np.int64), device=dev)
    sig_t = torch.tensor(sig, dtype=torch.float32, device=dev)
    meta_d = meta.to(dev, non_blocking=True)
    if out is None:
        degraded = torch.empty((b, 3, patch, patch), dtype=torch.float32, device=dev)
        clean = torch.empty_like(degraded)
    else:
        degraded, clean = out
    check(lib.pir_crop_augment_u8(images.data_ptr(), meta_d.data_ptr(), sig_t.data_ptr(), keys_t.data_ptr(),
                                  degraded.data_ptr(), clean.data_ptr(), b, patch, _stream()), "pir_crop_augment_u8")
    return degraded, clean


def crop_augment_host(item: dict, patch: int):
    """Host statement of the same item for the tests: numpy crop + utils/image_utils.py data_augmentation + ToTensor,
    and for unpaired samples the uint8-domain noise with the counter generator (element order CHW, as the device)."""
    t, l = item["top"], item["left"]
    cl = augment(item["clean"][t:t + patch, l:l + patch], item["mode"])
    clean = np.ascontiguousarray(cl.transpose(2, 0, 1))
    if item["degraded"] is not None:
        dg = augment(item["degraded"][t:t + patch, l:l + patch], item["mode"])
        deg = np.ascontiguousarray(dg.transpose(2, 0, 1))
    else:
        noise = W.normal01("noise", clean.size, item["noise_seed"]).reshape(clean.shape).astype(np.float64)
        deg = np.clip(clean.astype(np.float64) + noise * SIGMA_OF_DE_ID[item["de_id"]], 0, 255).astype(np.uint8)
    return deg.astype(np.float32) / np.float32(255.0), clean.astype(np.float32) / np.float32(255.0)


class RaggedDevicePrefetcher:
    """DevicePrefetcher for PromptTrainSet batches: the ragged uint8 buffer and its table go host -> device on a side
    stream from pinned memory, then ONE kernel cuts, augments, converts and degrades the patches there - one batch
    ahead of the step that consumes it."""

    def __init__(self, loader, device, patch: int):
        self.loader, self.device, self.patch = loader, device, patch
        self.stream = torch.cuda.Stream(device)

    def _stage(self, batch):
        with torch.cuda.stream(self.stream):
            images = batch["images"].to(self.device, non_blocking=True)
            degrad, clean = crop_augment_gpu(images, batch["meta"], batch["de_id"].tolist(), batch["noise_seed"].tolist(),
                                             self.patch)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return degrad, clean, ready, images

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            degrad, clean, ready, images = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            degrad.record_stream(cur)
            clean.record_stream(cur)
            del images
            yield degrad, clean


class DerainDehazeTestSet:
    """utils/dataset_utils.py:228-301 DerainDehazeDataset: <path>input/* with ground truth <path>target/<same name>
    (derain) or <path>target/<id>.png for input <id>_*.* (dehaze); crop_img(base=16); items (name, degraded, clean)
    uint8 HWC."""

    def __init__(self, derain_path: str, dehaze_path: str, task: str = "derain"):
        self.derain_path, self.dehaze_path = derain_path, dehaze_path
        self.set_dataset(task)

    def set_dataset(self, task: str):
        self.task_idx = {'derain': 0, 'dehaze': 1}[task]
        root = (self.derain_path if self.task_idx == 0 else self.dehaze_path) + 'input/'
        self.ids = [root + n for n in sorted(os.listdir(root))]

    def gt_path(self, degraded_name: str) -> str:
        if self.task_idx == 0:
            return degraded_name.replace("input", "target")
        return degraded_name.split("input")[0] + 'target/' + degraded_name.split('/')[-1].split('_')[0] + '.png'

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, idx):
        from PIL import Image

        p = self.ids[idx]
        deg = crop_img(np.array(Image.open(p).convert('RGB')), base=16)
        clean = crop_img(np.array(Image.open(self.gt_path(p)).convert('RGB')), base=16)
        return p.split('/')[-1][:-4], deg, clean

#!/usr/bin/env python3
"""Evaluation on the MI355X path: the reference's test.py (:84-164, :170-256).

    python evaluate.py --mode 0 --denoise_path test/denoise/ --ckpt_name model.ckpt     # bsd68, sigma 15 / 25 / 50
    python evaluate.py --mode 1 --derain_path test/derain/                              # Rain100L/input + target
    python evaluate.py --mode 2 --dehaze_path test/dehaze/                              # input + target (SOTS)
    python evaluate.py --mode 3 ...                                                     # all of the above
    python evaluate.py --synthetic 8                          # no data: deterministic synthetic images (mode 0)

Mode 0, for sigma in 15, 25, 50: add uint8-domain Gaussian noise (np.random.seed(0) as at test.py:183, then
utils/dataset_utils.py:195-198), mirror-pad to (H//64+1)*64 (test.py:100-104), restore, crop, PSNR with
data_range 1 on the clipped images (utils/val_utils.py:50-66).  Modes 1 / 2 (test.py:118-164): paired sets read with
DerainDehazeDataset's rules (utils/dataset_utils.py:228-301), same padding and PSNR.  SSIM / NIQE are not computed
(skimage is not a dependency of this path).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def crop_img(img, base=16):
    """utils/image_utils.py crop_img: centre crop to multiples of `base`."""
    h, w = img.shape[:2]
    ch, cw = h % base, w % base
    return img[ch // 2:h - ch + ch // 2, cw // 2:w - cw + cw // 2]


def load_set(opt):
    if opt.synthetic:
        from promptir_amd import weights as W

        imgs = []
        for i in range(opt.synthetic):
            clean = W.synthetic_clean(1, 160 + 16 * (i % 3), 208 - 16 * (i % 2), seed=900 + i)[0]
            imgs.append((f"synthetic_{i:03d}", np.floor(clean.transpose(1, 2, 0) * 255.0).astype(np.uint8)))
        return imgs
    from PIL import Image

    names = sorted(os.listdir(opt.denoise_path))
    return [(n.split('.')[0], crop_img(np.array(Image.open(os.path.join(opt.denoise_path, n)).convert('RGB')), 16))
            for n in names]


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--cuda', type=int, default=0)
    p.add_argument('--mode', type=int, default=0, help='0 for denoise, 1 for derain, 2 for dehaze, 3 for all-in-one (test.py:172-173)')
    p.add_argument('--denoise_path', type=str, default="test/denoise/")
    p.add_argument('--derain_path', type=str, default="test/derain/")
    p.add_argument('--dehaze_path', type=str, default="test/dehaze/")
    p.add_argument('--output_path', type=str, default="output/")
    p.add_argument('--ckpt_name', type=str, default="model.ckpt")
    p.add_argument('--synthetic', type=int, default=0, help='evaluate on N synthetic images instead of --denoise_path')
    p.add_argument('--save', action='store_true', help='write restored PNGs like the reference')
    opt = p.parse_args()
    if opt.mode not in (0, 1, 2, 3):
        raise SystemExit("--mode: 0 denoise, 1 derain, 2 dehaze, 3 all-in-one")
    if not torch.cuda.is_available():
        raise SystemExit("evaluate.py needs a ROCm device (no CPU fallback)")

    from net.model import PromptIR
    from promptir_amd.tile import mirror_pad_64, psnr
    from promptir_amd.train import load_checkpoint_file, load_lightning_checkpoint

    torch.cuda.set_device(opt.cuda)
    dev = torch.device("cuda", opt.cuda)
    net = PromptIR(decoder=True)
    ckpt_path = os.path.join("ckpt", opt.ckpt_name)
    if os.path.exists(ckpt_path):
        load_lightning_checkpoint(net, load_checkpoint_file(ckpt_path))
    else:
        print(f"[evaluate] {ckpt_path} not found: randomly initialised weights (PSNR is then meaningless)")
    net.to(dev).eval()
    np.random.seed(0)
    to_t = lambda a: torch.from_numpy(a.astype(np.float32).transpose(2, 0, 1) / 255.0)[None]

    def paired(task):
        """test_Derain_Dehaze (test.py:118-164)"""
        from promptir_amd.data import DerainDehazeTestSet

        derain_root = os.path.join(opt.derain_path, "Rain100L/")     # derain_splits, test.py:192
        dset = DerainDehazeTestSet(derain_root, opt.dehaze_path, task)
        vals = []
        for i in range(len(dset)):
            name, deg, clean = dset[i]
            x, t = to_t(deg).to(dev), to_t(clean).to(dev)
            with torch.no_grad():
                padded, h, w = mirror_pad_64(x)
                restored = net(padded)[:, :, :h, :w]
            vals.append(psnr(restored, t))
            if opt.save:
                from demo import save_image

                out_dir = os.path.join(opt.output_path, task)
                os.makedirs(out_dir, exist_ok=True)
                save_image(restored, os.path.join(out_dir, name + '.png'))
        print("%s PSNR: %.2f over %d images" % (task, sum(vals) / max(len(vals), 1), len(vals)))

    if opt.mode == 1:
        print('Start testing rain streak removal...')
        return paired("derain")
    if opt.mode == 2:
        print('Start testing SOTS...')
        return paired("dehaze")
    if opt.mode == 0 and not opt.synthetic:
        opt.denoise_path = os.path.join(opt.denoise_path, "bsd68/") if os.path.isdir(os.path.join(opt.denoise_path, "bsd68")) \
            else opt.denoise_path                                       # denoise_splits, test.py:191,197-198
    images = load_set(opt)
    for sigma in (15, 25, 50):
        vals = []
        for name, clean in images:
            noisy = np.clip(clean + np.random.randn(*clean.shape) * sigma, 0, 255).astype(np.uint8)
            x, t = to_t(noisy).to(dev), to_t(clean).to(dev)
            with torch.no_grad():
                padded, h, w = mirror_pad_64(x)
                restored = net(padded)[:, :, :h, :w]
            vals.append(psnr(restored, t))
            if opt.save:
                from demo import save_image

                out_dir = os.path.join(opt.output_path, 'denoise', str(sigma))
                os.makedirs(out_dir, exist_ok=True)
                save_image(restored, os.path.join(out_dir, name + '.png'))
        print("Denoise sigma=%d: psnr: %.2f over %d images" % (sigma, sum(vals) / len(vals), len(vals)))
    if opt.mode == 3:
        paired("derain")
        paired("dehaze")


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Denoising evaluation on the MI355X path; the `--mode 0` branch of the reference's test.py (:84-116, :170-219).

    python evaluate.py --denoise_path test/denoise/bsd68/ --ckpt_name model.ckpt
    python evaluate.py --synthetic 8                          # no data: deterministic synthetic images

For sigma in 15, 25, 50: add uint8-domain Gaussian noise (np.random.seed(0) as at test.py:183, then
utils/dataset_utils.py:195-198), mirror-pad to (H//64+1)*64 (test.py:100-104), restore, crop, PSNR with
data_range 1 on the clipped images (utils/val_utils.py:50-66).  SSIM / NIQE are not computed (skimage is not a
dependency of this path); derain / dehaze sets (modes 1-3) need their paired datasets and are out of scope.
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
    p.add_argument('--mode', type=int, default=0, help='0 for denoise (the only mode built)')
    p.add_argument('--denoise_path', type=str, default="test/denoise/")
    p.add_argument('--output_path', type=str, default="output/")
    p.add_argument('--ckpt_name', type=str, default="model.ckpt")
    p.add_argument('--synthetic', type=int, default=0, help='evaluate on N synthetic images instead of --denoise_path')
    p.add_argument('--save', action='store_true', help='write restored PNGs like the reference')
    opt = p.parse_args()
    if opt.mode != 0:
        raise SystemExit("only --mode 0 (denoise) is built")
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
    images = load_set(opt)
    to_t = lambda a: torch.from_numpy(a.astype(np.float32).transpose(2, 0, 1) / 255.0)[None]
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


if __name__ == '__main__':
    main()

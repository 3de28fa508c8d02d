#!/usr/bin/env python3
"""Restore images with the MI355X PromptIR path; same flags as the reference's demo.py (:79-92).

    python demo.py --test_path test/demo/ --output_path output/demo/ --ckpt_name model.ckpt --tile True

`--tile` keeps the reference's argparse quirk (type=bool: any non-empty string is True, demo.py:89).
Tiles are restored as one batch on the GPU (promptir_amd/tile.py); output PNGs use the reference's
`clip(x*255).astype(uint8)` truncation (utils/image_io.py:375-392).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def load_image(path):
    from PIL import Image

    return torch.from_numpy(np.array(Image.open(path).convert('RGB')).astype(np.float32).transpose(2, 0, 1) / 255.0)[None]


def save_image(t, path):
    from PIL import Image

    arr = np.clip(t.detach().cpu().numpy()[0].transpose(1, 2, 0) * 255.0, 0, 255).astype(np.uint8)
    Image.fromarray(arr).save(path)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--cuda', type=int, default=0)
    p.add_argument('--mode', type=int, default=3)
    p.add_argument('--test_path', type=str, default="test/demo/")
    p.add_argument('--output_path', type=str, default="output/demo/")
    p.add_argument('--ckpt_name', type=str, default="model.ckpt")
    p.add_argument('--tile', type=bool, default=False)
    p.add_argument('--tile_size', type=int, default=128)
    p.add_argument('--tile_overlap', type=int, default=32)
    opt = p.parse_args()

    from net.model import PromptIR
    from promptir_amd.tile import pad_input, tile_eval
    from promptir_amd.train import load_checkpoint_file, load_lightning_checkpoint

    if not torch.cuda.is_available():
        raise SystemExit("demo.py needs a ROCm device (no CPU fallback)")
    torch.cuda.set_device(opt.cuda)
    dev = torch.device("cuda", opt.cuda)
    net = PromptIR(decoder=True)
    ckpt_path = os.path.join("ckpt", opt.ckpt_name)
    if os.path.exists(ckpt_path):
        load_lightning_checkpoint(net, load_checkpoint_file(ckpt_path))
    else:
        print(f"[demo] {ckpt_path} not found: running with randomly initialised weights")
    net.to(dev).eval()
    os.makedirs(opt.output_path, exist_ok=True)
    paths = [opt.test_path] if os.path.isfile(opt.test_path) else sorted(
        os.path.join(opt.test_path, f) for f in os.listdir(opt.test_path))
    with torch.no_grad():
        for path in paths:
            x = load_image(path).to(dev)
            if opt.tile is False:
                restored = net(x)                                  # H, W must be multiples of 8, as in the reference
            else:
                x, h, w = pad_input(x)
                restored = tile_eval(net, x, tile=opt.tile_size, tile_overlap=opt.tile_overlap, crop=(h, w))
            save_image(restored, os.path.join(opt.output_path, os.path.basename(path).split('.')[0] + '.png'))
            print("[demo]", path, tuple(restored.shape))


if __name__ == '__main__':
    main()

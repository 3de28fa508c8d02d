#!/usr/bin/env python3
"""Train driver for the MI355X PromptIR path (replaces the reference's Lightning train.py).

    python train.py --epochs 2 --batch_size 8 --synthetic 64                     # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --num_gpus 8 ...

Same knobs as the reference's options.py (:3-39) where they apply; same recipe as train.py:28-56 — L1 loss,
AdamW(lr 2e-4, torch defaults), LinearWarmupCosineAnnealingLR(15, 150) stepped per EPOCH with the closed
form (so lr == 0 during epoch 0, as in the reference), one checkpoint per epoch whose `state_dict` keys are
`net.<PromptIR key>` like Lightning's.  Data: the reference's all-in-one folder layout (--data_file_dir with
--denoise_dir / --derain_dir / --dehaze_dir) if present, else --denoise_dir alone, else a deterministic synthetic set.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--model', type=str, default='promptir')
    p.add_argument('--cuda', type=int, default=0)
    p.add_argument('--epochs', type=int, default=120)
    p.add_argument('--batch_size', type=int, default=6, help="Batch size to use per GPU")
    p.add_argument('--lr', type=float, default=2e-4, help='accepted for compatibility; the reference ignores it (train.py:53)')
    p.add_argument('--de_type', nargs='+', default=['denoise_15', 'denoise_25', 'denoise_50', 'derain', 'dehaze'])
    p.add_argument('--patch_size', type=int, default=128)
    p.add_argument('--num_workers', type=int, default=16)
    p.add_argument('--data_file_dir', type=str, default='data_dir/', help='list files: noisy/denoise.txt, rainy/rainTrain.txt, hazy/hazy_outside.txt')
    p.add_argument('--denoise_dir', type=str, default='data/Train/Denoise/')
    p.add_argument('--derain_dir', type=str, default='data/Train/Derain/')
    p.add_argument('--dehaze_dir', type=str, default='data/Train/Dehaze/')
    p.add_argument('--ckpt_dir', type=str, default='train_ckpt')
    p.add_argument('--num_gpus', type=int, default=1)
    p.add_argument('--synthetic', type=int, default=0, help='use N synthetic samples per epoch (default when no data dir)')
    p.add_argument('--resume', type=str, default=None, help='checkpoint to resume from')
    p.add_argument('--max_steps', type=int, default=0, help='stop after this many optimiser steps (0 = no limit)')
    p.add_argument('--start_epoch', type=int, default=None)
    p.add_argument('--gpu_degrade', type=int, default=1,
                   help='synthetic data: add the uint8-domain noise on the device (pir_degrade_gaussian) instead of on the host')
    return p.parse_args()


def main():
    opt = parse()
    if opt.model != 'promptir':
        raise SystemExit("only --model promptir is built (SURVEY §2: sibling networks are out of scope)")
    from net.model import PromptIR
    from promptir_amd import data as D
    from promptir_amd import ops
    from promptir_amd.train import (DataParallelTrainer, init_distributed, load_checkpoint_file, load_lightning_checkpoint,
                                    lightning_epoch_lr)

    rank, local, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs a ROCm device (no CPU fallback)")
    device = torch.device("cuda", local)
    de_ids = [i for i, n in enumerate(['denoise_15', 'denoise_25', 'denoise_50']) if n in opt.de_type]
    gpu_degrade, ragged = False, False
    if os.path.isdir(opt.data_file_dir) and not opt.synthetic:
        # the reference's all-in-one layout (options.py:20-27, utils/dataset_utils.py:15-175): whole decoded images go to
        # the device, where ONE kernel crops, augments, converts and degrades them (RaggedDevicePrefetcher)
        dataset = D.PromptTrainSet(opt.data_file_dir, opt.denoise_dir, opt.derain_dir, opt.dehaze_dir, opt.de_type, opt.patch_size)
        ragged = True
    elif os.path.isdir(opt.denoise_dir) and not opt.synthetic:
        dataset = D.DenoiseFolderTrainSet(opt.denoise_dir, opt.patch_size, de_ids or [0, 1, 2])
    elif opt.gpu_degrade:
        dataset = D.CleanPatchSet(opt.synthetic or 64 * opt.batch_size * world, opt.patch_size, de_ids or [0, 1, 2])
        gpu_degrade = True
    else:
        dataset = D.SyntheticTrainSet(opt.synthetic or 64 * opt.batch_size * world, opt.patch_size, de_ids or [0, 1, 2])
    if rank == 0:
        pg = torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"
        print(f"[train] {type(dataset).__name__} with {len(dataset)} samples, world {world}, batch {opt.batch_size}/GPU, "
              f"process group {pg}")

    net = PromptIR(decoder=True)
    start_epoch, ckpt = 0, None
    if opt.resume:
        ckpt = load_checkpoint_file(opt.resume)
        load_lightning_checkpoint(net, ckpt)
        start_epoch = int(ckpt.get("epoch", -1)) + 1
    if opt.start_epoch is not None:
        start_epoch = opt.start_epoch
    net.to(device)
    trainer = DataParallelTrainer(net)
    if ckpt is not None and ckpt.get("optimizer_states"):
        trainer.opt.load_state_dict(ckpt["optimizer_states"][0], net)   # flat or torch/Lightning AdamW layout
    os.makedirs(opt.ckpt_dir, exist_ok=True)

    steps = 0
    for epoch in range(start_epoch, opt.epochs):
        lr = lightning_epoch_lr(epoch)                    # utils/schedulers.py:332-346 via train.py:48-50
        idx = D.shard_indices(len(dataset), rank, world, epoch)
        # DataLoader(batch_size, drop_last=True, num_workers, pin_memory=True) as reference train.py:336, with the
        # rank's DistributedSampler-style shard as its sampler; the next batch is copied (and degraded) on a side
        # stream while the current step runs, and the logged loss is accumulated on the device: no host sync per step.
        loader = torch.utils.data.DataLoader(dataset, batch_size=opt.batch_size, sampler=idx, drop_last=True,
                                             num_workers=min(opt.num_workers, os.cpu_count() or 1), pin_memory=True,
                                             persistent_workers=False, collate_fn=D.ragged_collate if ragged else None)
        t0, nb = time.time(), 0
        running = torch.zeros((), dtype=torch.float32, device=device)
        batches = D.RaggedDevicePrefetcher(loader, device, opt.patch_size) if ragged else D.DevicePrefetcher(loader, device, gpu_degrade)
        for degrad, clean in batches:
            loss = trainer.train_step(degrad, clean, lr=lr)
            ops.add_(running, loss)
            nb += 1
            steps += 1
            if opt.max_steps and steps >= opt.max_steps:
                break
        running = float(running)      # the epoch's only device -> host synchronisation
        if rank == 0:
            dt = time.time() - t0
            print(f"[train] epoch {epoch} lr {lr:.3e} train_loss {running / max(nb, 1):.5f} "
                  f"{nb * opt.batch_size * world / max(dt, 1e-9):.1f} patches/s")
            path = os.path.join(opt.ckpt_dir, f"epoch={epoch}-step={trainer.opt.steps}.ckpt")
            torch.save(trainer.checkpoint(epoch), path)   # scheduler block: the state AFTER this epoch's scheduler step
        if opt.max_steps and steps >= opt.max_steps:
            break
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

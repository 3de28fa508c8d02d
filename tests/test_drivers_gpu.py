"""Next-row callers on the GPU: train CLI (checkpoint interchange, resume), GPU-side degradation."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from promptir_amd import weights as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_degrade_gaussian_gpu_matches_host_generator():
    from promptir_amd.data import degrade_gaussian_gpu

    sig = [15, 25, 50, 25]
    deg, clean = W.synthetic_pair(4, 64, 64, sigma=sig, seed=5)
    out = degrade_gaussian_gpu(torch.from_numpy(clean).cuda(), sig, seed=5).cpu().numpy()
    diff = np.abs(out - deg)
    # identical generator; a pixel may differ by one grey level only where libm rounding moves the value
    # across an integer boundary
    assert float(diff.max()) <= 1.0 / 255 + 1e-7
    assert float((diff > 0).mean()) <= 1e-4
    assert abs(float((out - clean).std()) - float((deg - clean).std())) < 1e-4


def test_train_cli_checkpoint_and_resume(tmp_path):
    ck = tmp_path / "ck"
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--epochs", "2", "--batch_size", "2", "--synthetic", "4",
           "--patch_size", "64", "--ckpt_dir", str(ck), "--start_epoch", "1", "--max_steps", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    files = sorted(os.listdir(ck))
    assert files and files[0].startswith("epoch=1-step=")
    ckpt = torch.load(ck / files[0], map_location="cpu")
    assert all(k.startswith("net.") for k in ckpt["state_dict"]) and len(ckpt["state_dict"]) == 548
    # interchange: loads into a fresh module (the reference's load_from_checkpoint contract)
    from net.model import PromptIR
    from promptir_amd.train import load_lightning_checkpoint

    net = PromptIR(decoder=True)
    load_lightning_checkpoint(net, ckpt)
    cmd2 = cmd[:2] + ["--epochs", "3", "--batch_size", "2", "--synthetic", "4", "--patch_size", "64", "--ckpt_dir", str(ck),
                      "--resume", str(ck / files[0]), "--max_steps", "1"]
    out2 = subprocess.run(cmd2, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert "epoch 2" in out2.stdout


def test_evaluate_cli_synthetic():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "evaluate.py"), "--synthetic", "2"], capture_output=True,
                         text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("Denoise sigma=")]
    assert len(lines) == 3 and all("psnr:" in l for l in lines)


def _run(cmd, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    return subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=timeout, env=env)


def test_bench_runs_under_an_rccl_process_group():
    """The N>1 code path on the one GPU there is: `PIR_FORCE_PG=1` makes a world-size-1 job initialise the RCCL
    (`nccl`) process group, broadcast the flat parameters, capture the hipGraph beside the RCCL watchdog thread,
    all-reduce the flat gradient over RCCL every step, and destroy the group (reference train.py:336-341: DDP over
    NCCL).  Runs in a fresh child process."""
    import json

    out = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                "--batch", "8", "--no-cpu-baseline"],
               {"PIR_FORCE_PG": "1", "PIR_STAGED": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0",
                "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["config"]["process_group"] == "nccl"
    assert rec["config"]["execution"].startswith("hipGraph=1")
    assert "overlapped" in rec["config"]["execution"]      # the N>1 mode: three segment graphs, async RCCL all-reduces
    assert rec["config5"] is None                          # --batch 8 IS config 5
    assert rec["config"]["step1_loss_check"]["ok"] is True


def test_bench_with_two_ranks_sharing_the_gpu():
    """bench.py exactly as the driver launches it for N=2 (torch.distributed.run, one process per rank), rehearsed on the one
    GPU there is: both ranks use cuda:0 (PIR_SHARE_GPU=1) and, because RCCL refuses two ranks on one device, the gradient
    all-reduces go over gloo (PIR_DIST_BACKEND=gloo).  Exercises what a 1-rank group cannot: rank-sharded batches, the
    barrier + max-over-ranks timing, the segmented backward with a real exchange between replicas, every rank taking part
    in every collective (including the instrumented step), the config-5 leg, rank 0 printing the one JSON line."""
    import json

    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                "--warmup", "1", "--batch", "4"],
               {"PIR_SHARE_GPU": "1", "PIR_DIST_BACKEND": "gloo"}, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["process_group"] == "gloo" and rec["config"]["global_batch"] == 8
    assert rec["scaling"] == "weak" and rec["value"] > 0 and rec["per_gpu_value"] * 2 == pytest.approx(rec["value"], rel=2e-2)
    assert "overlapped" in rec["config"]["execution"]
    assert rec["config5"] is not None and rec["config5"]["global_batch"] == 16
    assert rec["roofline"]["achieved"] > 0 and rec["cpu_baseline"] is None     # cpu_baseline: N=1 only


def test_train_cli_under_an_rccl_process_group(tmp_path):
    out = _run([sys.executable, os.path.join(ROOT, "train.py"), "--epochs", "3", "--batch_size", "4", "--synthetic", "8",
                "--patch_size", "64", "--ckpt_dir", str(tmp_path / "ck"), "--start_epoch", "2", "--max_steps", "2"],
               {"PIR_FORCE_PG": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534", "RANK": "0", "WORLD_SIZE": "1",
                "LOCAL_RANK": "0"})
    assert out.returncode == 0, out.stderr[-3000:]
    assert "process group nccl" in out.stdout


def test_gpu_degradation_pipeline_equals_the_host_pipeline():
    """train.py's default input path (CleanPatchSet -> DataLoader -> DevicePrefetcher with pir_degrade_gaussian on a side
    stream) yields the batches of the host path (SyntheticTrainSet, utils/degradation_utils.py:21-27 semantics)."""
    from promptir_amd import data as D

    dev = torch.device("cuda:0")
    idx = [5, 0, 7, 2, 3, 9]
    host = D.SyntheticTrainSet(12, 64)
    loader = torch.utils.data.DataLoader(D.CleanPatchSet(12, 64), batch_size=3, sampler=idx, drop_last=True, num_workers=2,
                                         pin_memory=True)
    got = list(D.DevicePrefetcher(loader, dev, gpu_degrade=True))
    assert len(got) == 2
    for k, (deg, clean) in enumerate(got):
        ref_deg = torch.stack([host[i][1] for i in idx[3 * k:3 * k + 3]])
        ref_clean = torch.stack([host[i][2] for i in idx[3 * k:3 * k + 3]])
        assert torch.equal(clean.cpu(), ref_clean)
        diff = (deg.cpu() - ref_deg).abs()
        assert float(diff.max()) <= 1.0 / 255 + 1e-7 and float((diff > 0).float().mean()) <= 1e-4

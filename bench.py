#!/usr/bin/env python3
"""Headline benchmark: 128x128 patches/s (fwd + L1 + bwd + grad all-reduce + AdamW) of PromptIR(decoder=True).

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

One JSON line on rank 0 (contract in the task statement).  `value` is the whole-job patches/s with
inputs resident in HBM.  `roofline` describes the dominant kernel family (the kernels behind pir_gemm_nn - tiled gemm_nn_x3_kernel and the
persistent gemm_nn_bst_kernel / gemm_nn_res_kernel / gemm_nn_cst_kernel: every 1x1 convolution and its input gradient - also where
the call carries a LayerNorm on load or a LayerNorm backward in its store tail - and attn@v) measured live with HIP events on the launch
stream in one extra, instrumented step after the timed region.  `cpu_baseline` is the CPU oracle
(oracle/promptir_ref.py, PyTorch fp32 on the host cores) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide)
X3_PASSES = 6                   # bf16 MFMAs issued per fp32-class product block (hi/mid/lo split, gemm_common.h)
PEAK_HBM_GBS = 8000.0


def build_batch(batch, size, rank, device):
    from promptir_amd import weights as W

    sigmas = [(15, 25, 50)[i % 3] for i in range(batch)]          # all-in-one denoise mix surrogate
    degraded, clean = W.synthetic_pair(batch, size, size, sigma=sigmas, seed=100 + rank)
    return torch.from_numpy(degraded).to(device), torch.from_numpy(clean).to(device)


def build_model(device, seed=0):
    from net.model import PromptIR
    from promptir_amd import weights as W

    net = PromptIR(decoder=True)
    sd = {k: torch.from_numpy(W.make_tensor(k, tuple(v.shape), seed)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    return net.to(device), sd


def _usable_cpus():
    """Cores this process may really use: the affinity mask, cut by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sd, size, sample_batch=2, budget_s=30.0):
    """fwd + L1 + bwd of the CPU oracle on `sample_batch` patches of the same workload, swept over a few torch
    thread counts (an oversubscribed pool is several times slower than a well-sized one); the best is reported with
    its thread count.  Bounded: the sweep stops once `budget_s` seconds of CPU work have been spent."""
    from oracle import promptir_ref as O
    from promptir_amd import weights as W

    degraded, clean = W.synthetic_pair(sample_batch, size, size, sigma=[25, 50][:sample_batch] if sample_batch <= 2 else 25,
                                       seed=100)
    x, t = torch.from_numpy(degraded), torch.from_numpy(clean)
    usable, before = _usable_cpus(), torch.get_num_threads()
    tried, spent = [], 0.0
    for nt in sorted({min(n, usable) for n in (8, 16, 32, 64)}):
        if tried and spent + tried[-1][1] > budget_s:
            break
        torch.set_num_threads(nt)
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        t0 = time.perf_counter()
        loss = O.l1_loss(O.promptir_forward(params, x), t)
        loss.backward()
        dt = time.perf_counter() - t0
        tried.append((nt, dt))
        spent += dt
    torch.set_num_threads(before)
    best_nt, best_dt = min(tried, key=lambda r: r[1])
    return {"value": round(sample_batch / best_dt, 4), "unit": "patches/s", "cores": best_nt, "kind": "port",
            "sample": f"1 step fwd+L1+bwd, batch {sample_batch} x 3x{size}x{size}, PyTorch CPU fp32 oracle; best of torch "
                      f"threads {[n for n, _ in tried]} = {[round(d, 1) for _, d in tried]} s; {usable} usable of "
                      f"{os.cpu_count()} host cpus"}


def step1_loss_check(loss_value, batch, patch, rank):
    """The first forward of the timed configuration against the REAL reference's loss on the same synthetic batch
    (tests/golden/bench_step1_loss.json, oracle/make_golden.py benchloss).  A fast path with different results is not
    a result: a mismatch aborts the benchmark."""
    path = os.path.join(ROOT, "tests", "golden", "bench_step1_loss.json")
    key = f"b{batch}_rank{rank}"
    if patch != 128 or not os.path.exists(path):
        return None
    ref = json.load(open(path)).get(key)
    if ref is None:
        return None
    ok = abs(loss_value - ref) <= 2e-6
    if not ok:
        raise SystemExit(f"bench.py: step-1 loss {loss_value!r} differs from the reference's {ref!r} ({key})")
    return {"value": loss_value, "reference": ref, "tol": 2e-6, "ok": True}


def _time_calls(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def inference_leg(net, sd, device, steps, warmup, check_cpu=True):
    """BASELINE config 2: batch 8 x 3x128x128 sigma=25 forward (no_grad, hipGraph replay), patches/s; the first two
    outputs against the CPU oracle on the same inputs (max-abs and PSNR difference, north_star's parity bar)."""
    from promptir_amd import weights as W
    from promptir_amd.infer import GraphedForward

    degraded, clean = W.synthetic_pair(8, 128, 128, sigma=25, seed=200)
    x = torch.from_numpy(degraded).to(device)
    g = GraphedForward(net)
    y = g(x)
    dt = _time_calls(lambda: g(x), steps, warmup)
    leg = {"workload": "BASELINE config 2: inference forward, batch 8 x 3x128x128, sigma 25, no_grad, hipGraph replay",
           "value": round(8 / dt, 2), "unit": "patches/s", "ms_per_batch": round(dt * 1e3, 3)}
    if check_cpu:
        from oracle import promptir_ref as O

        with torch.no_grad():
            ref = O.promptir_forward(sd, torch.from_numpy(degraded[:2]))
        got, t = y[:2].cpu(), torch.from_numpy(clean[:2])
        leg["max_abs_err_vs_cpu_oracle"] = float((got - ref).abs().max())
        leg["psnr_db"] = round(O.psnr(got, t), 4)
        leg["psnr_db_cpu_oracle"] = round(O.psnr(ref, t), 4)
        leg["parity_ok"] = bool(leg["max_abs_err_vs_cpu_oracle"] <= 1e-4 and
                                abs(leg["psnr_db"] - leg["psnr_db_cpu_oracle"]) <= 1e-3)
    return leg


def tiled_leg(net, device, steps, warmup):
    """BASELINE config 4: demo.py's tiled path on 1x3x512x512, tile 128 / overlap 32 = 25 tiles restored as one batch
    (gather kernel, graph-replayed forward, blend kernel), ms per image."""
    from promptir_amd import weights as W
    from promptir_amd.infer import GraphedForward
    from promptir_amd.tile import tile_eval

    degraded, _ = W.synthetic_pair(1, 512, 512, sigma=25, seed=9)
    x = torch.from_numpy(degraded).to(device)
    g = GraphedForward(net)
    dt = _time_calls(lambda: tile_eval(g, x, tile=128, tile_overlap=32), steps, warmup)
    return {"workload": "BASELINE config 4: demo.py tiled path, 1x3x512x512, tile 128 overlap 32 (25 tiles, one batch)",
            "value": round(dt * 1e3, 3), "unit": "ms/image", "higher_is_better": False,
            "tiles_per_s": round(25 / dt, 2), "images_per_s": round(1 / dt, 3)}


def pipeline_leg(trainer, device, batch, patch, steps):
    """The timed step again with its inputs arriving through the input pipeline instead of sitting in HBM: host workers
    produce clean 8-bit patches (DataLoader, pinned memory), DevicePrefetcher copies batch k+1 and adds the uint8-domain
    noise on a side stream (pir_degrade_gaussian) while step k computes (train.py's default synthetic path).  PCIe
    and the loader are inside the timed region here - which is why this is a leg of its own and never `value`."""
    from promptir_amd import data as D

    workers = max(2, min(8, _usable_cpus() // 2))
    ds = D.CleanPatchSet(batch * (steps + 3), patch)
    loader = torch.utils.data.DataLoader(ds, batch_size=batch, num_workers=workers, pin_memory=True, drop_last=True)
    it = iter(D.DevicePrefetcher(loader, device, gpu_degrade=True))
    for _ in range(2):
        trainer.train_step(*next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_step(*next(it))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del it
    return {"workload": f"the headline step fed by DataLoader({workers} workers, pinned) + DevicePrefetcher + GPU-side degradation",
            "value": round(batch / dt, 3), "unit": "patches/s", "ms_per_step": round(dt * 1e3, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32,
                    help="patches per GPU of the headline line: 32 = BASELINE config 3 at N=1, the same per-GPU work at every "
                         "N (weak scaling); BASELINE config 5 (batch 8 per GPU) is measured beside it when N > 1")
    ap.add_argument("--config5", type=int, default=None,
                    help="0: skip the BASELINE config 5 leg (per-GPU batch 8 train step, reported under `config5`; default on)")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the inference (config 2) and tiled 512x512 (config 4) legs (rank 0, after the timed region)")
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from promptir_amd import ops
    from promptir_amd.train import DataParallelTrainer, init_distributed

    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the HIP path has no CPU fallback")
    device = torch.device("cuda", local)
    if args.batch == 32:
        baseline_cfg = "BASELINE config 3" if world == 1 else "BASELINE config 3's batch per GPU, data-parallel over %d GPUs" % world
    elif args.batch == 8 and world > 1:
        baseline_cfg = "BASELINE config 5"
    else:
        baseline_cfg = "non-BASELINE batch size"
    want_config5 = (args.batch != 8) if args.config5 is None else bool(args.config5)

    net, sd = build_model(device)
    trainer = DataParallelTrainer(net, lr=2e-4)
    x, t = build_batch(args.batch, args.patch, rank, device)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trainer.prepare(x, t)          # hipGraph capture of fwd+bwd for this batch shape (not a training step)
    # parity gate, outside the timed region: forward + loss + backward in the timed execution mode, no update
    loss_check = step1_loss_check(float(trainer.forward_backward(x, t)), args.batch, args.patch, rank)
    for _ in range(args.warmup):
        trainer.train_step(x, t)
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = trainer.train_step(x, t)
        marks[i + 1].record()      # on the stream the step is enqueued on: per-step device time
    sync()
    elapsed = time.perf_counter() - t0
    per_step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step_ms[len(per_step_ms) // 2] if len(per_step_ms) % 2 else \
        0.5 * (per_step_ms[len(per_step_ms) // 2 - 1] + per_step_ms[len(per_step_ms) // 2])
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # BASELINE config 5 (DDP, batch 8 per GPU) beside the headline line: same trainer, graph re-captured for the shape
    config5 = None
    if want_config5:
        x8, t8 = build_batch(8, args.patch, rank, device)
        trainer.prepare(x8, t8)
        for _ in range(args.warmup):
            trainer.train_step(x8, t8)
        sync()
        t5 = time.perf_counter()
        for _ in range(args.steps):
            trainer.train_step(x8, t8)
        sync()
        e5 = time.perf_counter() - t5
        if world > 1:
            tm = torch.tensor([e5], dtype=torch.float64, device=device)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            e5 = float(tm.item())
        config5 = {"workload": "BASELINE config 5: batch 8/GPU x 3x%dx%d, DDP over %d GPU(s)" % (args.patch, args.patch, world),
                   "global_batch": 8 * world, "value": round(8 * world * args.steps / e5, 3), "unit": "patches/s",
                   "ms_per_step": round(e5 / args.steps * 1e3, 3), "per_gpu_value": round(8 * args.steps / e5, 3)}
        trainer.prepare(x, t)      # back to the headline shape for the instrumented step below

    # N > 1 (VERDICT r3 #6b): the same steps in BOTH gradient-exchange modes, so that the first run on real xGMI decides the
    # default: "single" = one graph + one exposed all-reduce of the 142 MB flat gradient, "staged" = three segment graphs
    # with the all-reduce of each finished range overlapping the next segment.  Headline batch and config 5's batch 8.
    staged_ab = None
    if world > 1 or os.environ.get("PIR_BENCH_STAGED_AB") == "1":
        default_mode = trainer.staged
        staged_ab = {"default": "staged" if default_mode else "single"}
        shapes = [("batch%d" % args.batch, x, t)] + ([("batch8", *build_batch(8, args.patch, rank, device))] if args.batch != 8 else [])
        for mode in (False, True):
            if trainer.set_staged(mode) != mode:
                continue
            for tag, xb, tb in shapes:
                trainer.prepare(xb, tb)
                for _ in range(max(1, args.warmup)):
                    trainer.train_step(xb, tb)
                sync()
                ta = time.perf_counter()
                for _ in range(args.steps):
                    trainer.train_step(xb, tb)
                sync()
                ea = time.perf_counter() - ta
                if world > 1:
                    tm = torch.tensor([ea], dtype=torch.float64, device=device)
                    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                    ea = float(tm.item())
                staged_ab["%s_%s_ms" % ("staged" if mode else "single", tag)] = round(ea / args.steps * 1e3, 3)
        trainer.set_staged(default_mode)
        trainer.prepare(x, t)

    roofline, cpu = None, None
    # one extra instrumented step: HIP events around every C-ABI call on the launch stream.  Every rank runs it (the
    # step holds the gradient all-reduce, and the replicas must stay in step); rank 0 reports its own records.
    ops.lib.start_timing()
    trainer.train_step(x, t)
    recs = ops.lib.stop_timing()
    if rank == 0:
        fam = {}
        for name, sec, work, nbytes in recs:
            f = fam.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
            f[0] += 1; f[1] += sec; f[2] += work; f[3] += nbytes
            # this launch's own roofline time: the larger of its MFMA time (bf16x3 ceiling) and its HBM time
            f[4] += max(work / (PEAK_BF16_MFMA_TFLOPS / X3_PASSES * 1e12), nbytes / (PEAK_HBM_GBS * 1e9))

        def group(pred):
            tot = [0, 0.0, 0.0, 0.0, 0.0]
            for k, v in fam.items():
                if pred(k.split("@")[0], k.split("@")[1] if "@" in k else None):
                    tot = [a + b for a, b in zip(tot, v)]
            return tot

        # the dominant family: the bf16x3 "nn" GEMM kernels = every 1x1 convolution forward and input gradient, whether called
        # plain (pir_gemm_nn), with the LayerNorm applied on load (pir_ln_conv1x1_fwd) or with the LayerNorm backward in the
        # store tail (pir_conv1x1_dgrad_ln_bwd); flops and bytes are the GEMM's / the fused call's algorithmic ones
        calls, secs, flops, gemm_bytes, gemm_bound = group(
            lambda n, tag: n in ("pir_gemm_nn", "pir_ln_conv1x1_fwd", "pir_conv1x1_dgrad_ln_bwd"))
        total = sum(v[1] for v in fam.values())
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        x3 = ops.USE_X3 and os.environ.get("PIR_NN_X3", "1") != "0"
        # bf16x3: each fp32-class product costs 6 bf16 MFMA passes, so the ceiling for ALGORITHMIC flops is
        # bf16 peak / 6; `frac` is then the MFMA-pipe utilisation.  The fp32-MFMA figure is kept beside it.
        peak = PEAK_BF16_MFMA_TFLOPS / X3_PASSES if x3 else PEAK_F32_MFMA_TFLOPS

        def family(pred, bound):
            """One family of the instrumented step against its roofline: `hbm` - algorithmic bytes / event time / 8 TB/s;
            `mfma` - algorithmic flops / event time / the bf16x3 ceiling; both carry the per-launch max(MFMA, HBM) bound."""
            n, sec, fl, by, bnd = group(pred)
            if n == 0 or sec <= 0:
                return None
            out = {"bound": bound, "abi_calls": n, "ms": round(sec * 1e3, 3),
                   "frac_of_per_launch_bound": round(bnd / sec, 4)}
            if bound == "hbm":
                out.update(achieved=round(by / sec / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(by / sec / 1e9 / PEAK_HBM_GBS, 4))
            else:
                out.update(achieved=round(fl / sec / 1e12, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(fl / sec / 1e12 / peak, 4),
                           algorithmic_gbs=round(by / sec / 1e9, 1))
            return out

        STENCILS = ("pir_dwconv3x3_sumsq", "pir_dwconv3x3", "pir_dwconv3x3_gate", "pir_dwconv3x3_bwd", "pir_gdfn_dwconv_bwd")
        families = {
            # north_star: "stated fraction of HBM roofline for depthwise": algorithmic planes x 4 B
            "depthwise_stencils": family(lambda n, tag: n in STENCILS, "hbm"),
            **{"depthwise:" + k: family(lambda n, tag, k=k: n == k, "hbm") for k in STENCILS if any(f.split("@")[0] == k for f in fam)},
            # north_star: "... and of MFMA roofline for MDTA": q k^T, attn v (folded into project_out where the fold runs) and
            # their backward products dW_eff, dv, dq, dk - the calls ops.py files under the "mdta" tag
            "mdta_contractions": family(lambda n, tag: tag == "mdta" and n in ("pir_gemm_nn", "pir_gemm_nt", "pir_mdta_dqk"), "mfma"),
            # every weight gradient over the pixels that is not an MDTA product
            "weight_gradients": family(lambda n, tag: tag != "mdta" and n in ("pir_gemm_nt", "pir_gemm_nt_group", "pir_conv1x1_wgrad_ln", "pir_conv3x3_wgrad"), "mfma"),
            "dense_conv3x3": family(lambda n, tag: n in ("pir_conv3x3", "pir_conv3x3_x3"), "mfma"),
        }
        # HBM bytes from the committed PMC profile of this build (rocprofv3 cannot run inside the timed process):
        # per launch of the dominant family (FETCH_SIZE x its calibrated factor + WRITE_SIZE) and summed over the whole step
        traffic, traffic_note, step_bytes = None, None, None
        tpath = os.path.join(ROOT, "profiles", "r04_traffic.json")
        if not os.path.exists(tpath):
            tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if x3 and args.batch == 32 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            fam_t = tj["families"].get("pir_gemm_nn")
            if fam_t:
                fx = fam_t.get("fetch_factor", 2.0)
                traffic = round((fx * fam_t["fetch_kb_per_launch_raw"] + fam_t["write_kb_per_launch"]) * 1024)
                traffic_note = os.path.relpath(tpath, ROOT) + ": " + tj["source"] + "; " + tj["note"]
            step_bytes = tj.get("step_bytes")
        lpath = os.path.join(ROOT, "profiles", "r04_step_launches.json")
        launches = json.load(open(lpath)).get("batch%d" % args.batch) if os.path.exists(lpath) else None
        step_s = elapsed / args.steps
        roofline = {"kernel": ("gemm_nn_x3_kernel + gemm_nn_bst_kernel + gemm_nn_res_kernel + gemm_nn_cst_kernel" if x3 else "gemm_nn_kernel") +
                              " (every kernel behind pir_gemm_nn, pir_ln_conv1x1_fwd and pir_conv1x1_dgrad_ln_bwd)",
                    "bound": "mfma", "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_note": traffic_note,
                    "peak_note": "algorithmic fp32-class flops; peak = 2500 TF/s dense bf16 / 6 MFMA passes per product"
                                 if x3 else "fp32 MFMA peak",
                    "vs_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    # the family mixes MFMA-bound and HBM-bound shapes: sum over launches of max(flops / MFMA ceiling,
                    # algorithmic bytes / HBM peak) over the measured time, and the family's algorithmic byte rate
                    "frac_of_per_launch_bound": round(gemm_bound / secs, 4) if secs > 0 else None,
                    "algorithmic_gbs": round(gemm_bytes / secs / 1e9, 1) if secs > 0 else None, "hbm_peak_gbs": PEAK_HBM_GBS,
                    "launches_per_step": calls, "avg_launch_us": round(secs / max(calls, 1) * 1e6, 2),
                    "share_of_step_kernel_time": round(secs / total, 4) if total > 0 else None,
                    "families": {k: v for k, v in families.items() if v is not None},
                    # the whole step: HBM bytes of one step from the committed counter passes (sum over every kernel of
                    # calibrated FETCH_SIZE + WRITE_SIZE) over THIS run's step time, against the 8 TB/s peak
                    "bytes_per_step": step_bytes,
                    "step_frac_of_hbm": round(step_bytes / step_s / (PEAK_HBM_GBS * 1e9), 4) if step_bytes else None,
                    "step_launches": {"abi_calls_instrumented_step": len(recs),
                                      "kernel_launches_per_step": launches,
                                      "note": "kernel launches per graph step, share under 10 us and kernels outside the library: "
                                              "profiles/r04_step_launches.json (tools/step_kernels.py over a rocprofv3 kernel trace)"},
                    "families_ms": {k: round(v[1] * 1e3, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}}
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N=1 only
            cpu = cpu_baseline(sd, args.patch)
    inference, tiled, pipeline = None, None, None
    if world == 1 and not args.no_legs and args.patch == 128:
        pipeline = pipeline_leg(trainer, device, args.batch, args.patch, max(args.steps // 2, 5))
    if rank == 0 and world == 1 and not args.no_legs and args.patch == 128:
        # BASELINE configs 2 and 4 (replicas only at N > 1: no collective, so they are measured at N = 1), after the
        # timed region, on the weights the timed steps left behind (re-split below: the optimiser changed them)
        ops.refresh_split_weights()
        net.eval()
        cur = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        inference = inference_leg(net, cur, device, max(args.steps, 10), 3, check_cpu=not args.no_cpu_baseline)
        tiled = tiled_leg(net, device, max(args.steps // 2, 5), 2)
    sync()
    spread = None
    if world > 1 and os.environ.get("PIR_BENCH_PARAM_CHECK") == "1":
        # rehearsal check: the replicas must hold bit-identical parameters after the steps above
        ref_p = trainer.opt.param.clone()
        dist.broadcast(ref_p, src=0)
        d = (trainer.opt.param - ref_p).abs().max().reshape(1)
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        spread = float(d.item())

    if rank == 0:
        patches = args.batch * world * args.steps
        out = {
            "metric": "128x128 patches/sec (fwd+bwd train step, whole job)", "value": round(patches / elapsed, 3),
            "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_median": round(median_ms, 3), "ms_per_step_min": round(per_step_ms[0], 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (bf16x3 split-MFMA, f32 accumulate)" if ops.USE_X3 else "f32",
            "data": "synthetic",
            "config": {"workload": f"PromptIR(decoder=True) train step: fwd + L1 + bwd + grad all-reduce + AdamW, "
                                   f"batch {args.batch}/GPU x 3x{args.patch}x{args.patch}, all-in-one sigma mix "
                                   f"({baseline_cfg})",
                       "global_batch": args.batch * world, "patch": args.patch, "parallelism": f"dp{world}",
                       "execution": f"hipGraph={int(trainer.graph)}, part-batch streams={trainer._nparts(args.batch)}" +
                                    (", backward in 3 segments with overlapped gradient all-reduce" if trainer.staged else ""),
                       "process_group": dist.get_backend() if dist.is_initialized() else None,
                       "world": world, "device_index": local, "replica_param_spread": spread,
                       "staged_backward": bool(trainer.staged),
                       "gradient_ranges_bytes": [[4 * lo, 4 * hi] for lo, hi in (trainer.opt.stages or [])],
                       "params": 35592263, "final_loss": float(loss), "step1_loss_check": loss_check,
                       # everything that can change which kernels ran: module switches, PIR_* environment, tuning knobs
                       "kernel_selection": ops.effective_switches(), "staged_ab": staged_ab},
            "per_gpu_value": round(patches / elapsed / world, 3),
            "roofline": roofline, "cpu_baseline": cpu, "config5": config5, "inference": inference, "tiled_512": tiled,
            "input_pipeline": pipeline,
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

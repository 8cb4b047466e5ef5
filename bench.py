#!/usr/bin/env python3
"""Headline benchmark: data-parallel DiT-S/2 training step on synthetic 4x32x32 latents (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--batch B]

One process per GPU (RANK/LOCAL_RANK/WORLD_SIZE from torch.distributed.run).  A step = sample (t, eps) +
q-sample -> DiT forward -> v/eps MSE loss -> backward -> gradient all-reduce (RCCL) -> AdamW + cosine LR, on
inputs already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import subprocess  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODEL = "DiT-S/2"
# per-GPU batch: the config does not fix it (the reference yaml trains at 16); sweep on one MI355X:
# 16 -> 3.6k, 64 -> 8.9k, 128 -> 11.7k, 256 -> 13.2k, 512 -> 14.1k, 768 -> 14.9k images/s.  768 = 33 GB of activations, and
# M = 196608 token rows is a whole number of rounds of the one-workgroup-per-CU GEMM tiles (1024 x 192 rows, 768 x 256).
DEFAULT_BATCH = 768
# step GFLOP per image (3 x forward, no recompute)
STEP_GFLOP = {"DiT-S/2": 36.3, "DiT-B/2": 138.0, "DiT-L/2": 484.0, "DiT-XL/2": 711.7,
              "SDXL-UNet": {32: 1283.0, 128: 20284.0}}  # the UNet2DConditionModel shape the reference YAMLs build
DESC = {"DiT-S/2": "L12 D384 h6", "DiT-B/2": "L12 D768 h12", "DiT-L/2": "L24 D1024 h16", "DiT-XL/2": "L28 D1152 h16",
        "SDXL-UNet": "2.57 B parameters, 77x2048 text context + text_time conditioning"}
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def cpu_baseline(batch=16, warm=2, iters=5):  # B=16, median of >= 5 steps after 2 warm-ups (~5 s)
    """The oracle (this build's fp32 PyTorch restatement of the same model + loss) timed on the host cores."""
    from oracle import loss as OL
    from oracle.dit import DiTOracle
    from oracle.scheduler import EulerDiscreteScheduler

    torch.manual_seed(0)
    # threads = the CPUs this process may actually run on (the GPU box gives 16 per GPU; os.cpu_count() reports
    # the whole host and oversubscribing it stalls for minutes)
    try:
        n_thr = len(os.sched_getaffinity(0))
    except AttributeError:
        n_thr = os.cpu_count() or 1
    n_thr = max(1, min(n_thr, 16))
    torch.set_num_threads(n_thr)
    print(f"[bench] cpu_baseline: oracle DiT-S/2 fp32 on {n_thr} threads ...", file=sys.stderr, flush=True)
    model = DiTOracle(depth=12, hidden=384, heads=6, cond_dim=1280)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn_like(p) * 0.02)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-6, weight_decay=0.01)
    sched = EulerDiscreteScheduler.sdxl()
    x = torch.randn(batch, 4, 32, 32)
    pooled = torch.randn(batch, 1280)

    def step():
        noise = torch.randn_like(x)
        t = torch.randint(0, 1000, (batch,))
        noisy = OL.q_sample(x, noise, OL.sigmas_for_timesteps(sched, t))
        out = model(noisy, t, added_cond_kwargs={"text_embeds": pooled})[0]
        loss = ((out - noise) ** 2).flatten(1).mean(1).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(warm):
        step()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
        print(f"[bench] cpu step {ts[-1]:.2f} s", file=sys.stderr, flush=True)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(batch / med, 2), "unit": "images/s", "cores": n_thr, "kind": "port",
            "sample": f"oracle DiT-S/2 fp32 CPU, batch {batch}, median of {iters} steps (fwd+bwd+AdamW) after {warm} warm-ups"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start one CHILD process per rank (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in its environment) and wait.  The parent never initialises the GPU and never exec()s; rank 0's stdout
    (the one JSON line) is this process's stdout."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    # HSA_ENABLE_IPC_MODE_LEGACY=0 (kept if the caller already set it): RCCL's intra-node transport over xGMI maps the peers'
    # buffers through hipIpcGetMemHandle; this pool's host driver supports only the dmabuf IPC mode, and with the legacy mode
    # that call fails ("invalid argument") before the first collective.  The image exports it already -- the default here
    # only covers a shell that lost it.  It has no effect on gloo.
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:  # one rank failed: the others would wait in a collective forever
                    rc = rc or code
                    for q in procs:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            p.kill()
    return rc


# kernel families of the live profiler (include/uwu_hip.h UWU_PROF_*): (tag, name, bound)
FAMILIES = [(0, "gemm_fwd (qkv / proj / fc2 forward)", "mfma"), (1, "gemm_dgrad (input gradients)", "mfma"),
            (2, "gemm_wgrad (streaming weight gradients + split-K reduce)", "mfma"),
            (3, "gemm fc1 + bias + GELU", "mfma"), (4, "gemm fc2 dgrad + dGELU", "mfma"),
            (5, "attention forward (attn_fwd_p256 at T = 256 / head dim 64, else attn_fwd_mfma)", "mfma"),
            (6, "attention backward (attn_bwd_p256 at T = 256 / head dim 64, else attn_bwd_mfma [+ attn_bwd_dq_mfma])", "mfma"),
            (7, "add_ln_mod_fwd", "hbm"), (8, "add_ln_mod_bwd", "hbm")]
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md "Chip-level parameters" (spec; 6.3 TB/s is what a copy reaches)


def collect_families(L, lib, n_prof, kind, gemm_peak=None):
    """Per-family roofline entries from the live profiler: achieved = algorithmic FLOPs (or bytes) / measured duration.
    gemm_peak: the MFMA peak the GEMM families (tags 0-4) are priced against (fp8 mode: 2 x the bf16 peak; the attention
    kernels compute in bf16 in every mode)."""
    out = []
    gemm_peak = gemm_peak or PEAK_BF16_TFLOPS
    for tag, name, bound in FAMILIES:
        ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        L.check(lib.uwu_prof_collect(tag, -1 if tag >= 5 else kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by),
                                     ctypes.byref(n)), "prof_collect")
        if not n.value or ms.value <= 0:
            continue
        sec = ms.value * 1e-3
        e = {"kernel": name, "bound": bound, "launches_per_step": n.value // n_prof,
             "avg_launch_us": round(ms.value * 1e3 / n.value, 2), "ms_per_step": round(ms.value / n_prof, 3),
             "tflops": round(fl.value / sec / 1e12, 1), "gbs": round(by.value / sec / 1e9, 1)}
        pk = gemm_peak if tag <= 4 else PEAK_BF16_TFLOPS
        e["frac"] = round(e["tflops"] / pk, 4) if bound == "mfma" else round(e["gbs"] / PEAK_HBM_GBS, 4)
        if bound == "mfma" and pk != PEAK_BF16_TFLOPS:
            e["peak"] = pk
        e["frac_hbm"] = round(e["gbs"] / PEAK_HBM_GBS, 4)
        out.append(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None,
                    help=f"per-GPU batch (weak scaling); default {DEFAULT_BATCH} (DiT), 12 / 48 (SDXL-UNet at 128 / 32 latents)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="bf16 (headline); fp8 = BASELINE config 5 (DiT block Linears on e4m3 / e5m2 operands, block-scaled MFMA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grad-checkpoint", action="store_true",
                    help="recompute in the backward (larger batches fit): DiT per transformer block inside the C++ driver, "
                         "SDXL-UNet per resnet / Transformer2D stack")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short runs of the other BASELINE configs (DiT-B/2, DiT-XL/2 fp8, SDXL-UNet 4x128x128) after the headline")
    ap.add_argument("--single-allreduce", action="store_true",
                    help="N > 1: ONE all-reduce of the whole flat gradient buffer after the backward (north_star's wording) instead "
                         "of the overlapped block-group + chunked exchange -- A/B switch for the scaling run")
    ap.add_argument("--no-sweep", action="store_true", help="skip the per-GPU batch sweep (16 / 64 / 256) after the timed region")
    ap.add_argument("--clip", type=float, default=0.0)
    ap.add_argument("--model", default=MODEL, choices=sorted(STEP_GFLOP),
                    help="denoiser; the headline metric (BASELINE.json) is DiT-S/2, the others are extra configs "
                         "(SDXL-UNet = BASELINE config 4: give --latent 128 --batch 6)")
    ap.add_argument("--latent", type=int, default=32, choices=[32, 128], help="latent side (DiT presets: 32)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) for real runs; gloo lets several ranks share ONE GPU to rehearse the "
                         "multi-process path on a 1-GPU box")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # nothing in this process has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: timing {world} rank(s)", file=sys.stderr)
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1) if args.backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from uwudiff_amd import lib as L
    from uwudiff_amd.dit import DiT
    from uwudiff_amd.gradsync import FlatGradSync
    from uwudiff_amd.objective import DiffusionLoss
    from uwudiff_amd.optim import FusedAdamW, cosine_lr
    from uwudiff_amd.scheduler import EulerDiscreteScheduler

    if args.batch is None:  # SDXL-UNet: ~12 GB of saved activations per 4x128x128 sample; 12 is 30.7 img/s, 6 is 24.8, 24 does not fit
        args.batch = DEFAULT_BATCH if args.model != "SDXL-UNet" else (12 if args.latent == 128 else 48)
    B = args.batch
    torch.manual_seed(1215 + rank)  # configs/demo_training_latent.yaml:1 + test_train.py:69 (seed + rank)
    # random (non-zero) weights everywhere: zero-initialised gates would make whole branches numerically dead
    unet = args.model == "SDXL-UNet"
    if unet:
        from uwudiff_amd.unet import UNet2DConditionModel

        model = UNet2DConditionModel.from_config("sdxl", compute_dtype=args.dtype, device=dev)  # (2.57 G initial weights drawn on the GPU)
        if args.grad_checkpoint:
            model.enable_gradient_checkpointing()
    else:
        model = DiT.from_config(args.model, cond_dim=1280, init="random", compute_dtype=args.dtype).to(dev)
        if args.grad_checkpoint:
            model.enable_gradient_checkpointing()
    if world > 1:  # identical replicas: broadcast rank 0's parameters
        dist.broadcast(model.flat.data, src=0)
        if hasattr(model, "refresh_shadow"):
            model.refresh_shadow()
    loss_fn = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("stabilityai/stable-diffusion-xl-base-1.0",
                                                                   subfolder="scheduler"))
    opt = FusedAdamW(model.parameters(), lr=1e-6, weight_decay=0.01, betas=(0.9, 0.999))
    # N > 1: block gradients are reduced while the backward still runs (--single-allreduce: one call after it)
    sync = FlatGradSync(world, single=args.single_allreduce).attach(model)
    S = args.latent
    step_gflop = STEP_GFLOP[args.model][S] if unet else STEP_GFLOP[args.model]
    pool_n = max(4096 if S == 32 else 64, 2 * B)
    # (synthetic inputs are drawn on the host and copied over: the kernel trace of this command then holds the library's kernels only)
    pool = torch.randn(pool_n, 4, S, S).to(dev)                # synthetic latents resident in HBM
    pooled = torch.randn(pool_n, 1280).to(dev)                 # synthetic pooled-text conditioning
    ctx = torch.randn(B, 77, 2048).to(dev) if unet else None   # synthetic text context (77 x 2048)
    time_ids = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B).to(dev) if unet else None
    model.flat.grad = torch.empty_like(model.flat.data)
    L.call("uwu_memset_zero", L.ptr(model.flat.grad), model.flat.grad.numel() * 4, L.stream())
    one = torch.ones((), dtype=torch.float32).to(dev)
    step_no = [0]

    def step(b=B, sy=None):
        sy = sy or sync
        i = step_no[0]
        off = (i * b) % (pool_n - b + 1)
        x, c = pool[off:off + b], pooled[off:off + b]
        # (the flat gradient buffer was zeroed by the AdamW kernel of the previous step: FusedAdamW.step(zero_grad=True))
        if unet:
            loss, _ = loss_fn(x, model, encoder_hidden_states=ctx[:b], added_cond_kwargs={"text_embeds": c, "time_ids": time_ids[:b]})
        else:
            loss, _ = loss_fn(x, model, added_cond_kwargs={"text_embeds": c})
        loss.backward(one)  # (a resident 1.0: autograd's own ones_like(loss) is a fill launch per step)
        chunks = sy.all_reduce(model.flat.grad)
        opt.param_groups[0]["lr"] = cosine_lr(1e-6, i, 100_000, 1e-7)
        if args.clip > 0:
            sy.wait_all()
            clip = opt.grad_norm_clip(args.clip, pre_scale=sy.pre_scale)
            opt.step(clip=clip, pre_scale=sy.pre_scale, zero_grad=True)
        else:
            opt.step(pre_scale=sy.pre_scale, chunks=chunks, before_chunk=sy.wait_chunk, zero_grad=True)
        step_no[0] += 1
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, **kw):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step(**kw)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = te.item()
        return el, loss

    if rank == 0:
        print(f"[bench] {args.model} {args.dtype} per-GPU batch {B} x {world} GPU(s): warm-up {args.warmup}, timing {args.steps} steps",
              file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    elapsed, loss = timed(args.steps)
    final_loss = float(loss.detach())
    if rank == 0:
        print(f"[bench] timed region done: {elapsed / args.steps * 1e3:.3f} ms/step", file=sys.stderr, flush=True)

    # ---- host side: wall time for the host to ENQUEUE one step (no sync inside); close to ms_per_step = launch-bound
    host_ms = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        host_ms.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    host_enqueue_ms = round(sorted(host_ms)[1], 3)

    # ---- N > 1: how much of the gradient exchange is NOT hidden = step time with the exchange - step time without it
    # (same kernels, the all-reduce calls skipped; replicas drift apart afterwards, which no longer matters)
    comm = None
    if world > 1:
        n_c = min(10, max(2, args.steps))
        t_sync, _ = timed(n_c)
        if hasattr(model, "set_grad_ready_hook"):
            model.set_grad_ready_hook(None)
        nosync = FlatGradSync(1)
        t_nosync, _ = timed(n_c, sy=nosync)
        comm = {"rccl_ranks": dist.get_world_size(), "backend": args.backend, "exchange_path": sync.path,
                "exchange": "one all-reduce after the backward" if args.single_allreduce else
                            "block groups reduced inside the backward + the rest in 32 MB chunks, AdamW per reduced slice",
                "ms_per_step_with_exchange": round(t_sync / n_c * 1e3, 3),
                "ms_per_step_without_exchange": round(t_nosync / n_c * 1e3, 3),
                "exposed_comm_ms": round((t_sync - t_nosync) / n_c * 1e3, 3),
                "grad_bytes": int(model.flat.numel() * 4)}
        sync_prof = nosync
    else:
        sync_prof = sync

    # ---- roofline: HIP events around every instrumented launch on the launch stream, over a replay of the same
    # steps (instrumentation kept out of the throughput window).  Dominant family = the bf16 MFMA GEMMs.
    roof = None
    if rank == 0:
        lib = L.load()
        n_prof = min(5, max(1, args.steps))
        L.check(lib.uwu_prof_enable(1), "prof_enable")
        for _ in range(n_prof):
            step(sy=sync_prof)
        torch.cuda.synchronize()
        L.check(lib.uwu_prof_enable(0), "prof_disable")
        ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        kind = 0 if args.dtype in ("bf16", "fp8") else 1
        L.check(lib.uwu_gemm_prof_collect(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), "prof_collect")
        if n.value:
            ach = fl.value / (ms.value * 1e-3) / 1e12
            peak = {"bf16": PEAK_BF16_TFLOPS, "fp8": 2 * PEAK_BF16_TFLOPS, "fp32": PEAK_BF16_TFLOPS / 16}[args.dtype]
            # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this
            # same command, gfx950 corrections applied by tools/summarize_pmc.py).  The file names the kernel source
            # it was measured on; a number from other kernels is not reported.
            traffic, traffic_src = None, None
            for tjn in ("r04_pmc_gemm_traffic.json", "r03_pmc_gemm_traffic.json"):
                tj = os.path.join(ROOT, "profiles", tjn)
                if traffic is None and args.dtype == "bf16" and B == DEFAULT_BATCH and args.model == MODEL and os.path.exists(tj) \
                        and not args.grad_checkpoint:
                    with open(tj) as f:
                        tjd = json.load(f)
                    if tjd.get("gemm_src_sha16") == src_sha16():
                        traffic = round(tjd["hbm_bytes_per_launch"])
                        traffic_src = f"profiles/{tjn} (separate --pmc passes over this command, same GEMM sources)"
            fam = ("fp8 block-scaled MFMA GEMM family (gemm_f8_kernel, v_mfma_scale_f32_16x16x128_f8f6f4: fwd + dgrad + split-K wgrad of the "
                   "block Linears; the few Linears outside the blocks stay bf16)" if args.dtype == "fp8" else
                   "bf16 MFMA GEMM family (gemm_as / gemm_wide / gemm_big / gemm_p8 / gemm_p8n / gemm_r3 / gemm_trw / gemm_tr kernels, "
                   "v_mfma_f32_16x16x32_bf16; fwd+dgrad+wgrad launches)")
            roof = {"bound": "mfma", "kernel": fam,
                    "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": n.value // n_prof,
                    "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
                    "gemm_ms_per_step": round(ms.value / n_prof, 3),
                    "kernels": collect_families(L, lib, n_prof, kind, gemm_peak=peak)}
    elif world > 1:
        for _ in range(min(5, max(1, args.steps))):
            step(sy=sync_prof)
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()

    # ---- per-GPU batch sweep (the reference yaml's 16, and 64 / 256), N = 1 only, short
    sweep = None
    if rank == 0 and world == 1 and not unet and not args.no_sweep:
        sweep = {}
        for b in (16, 64, 256):
            if b == B:
                continue
            for _ in range(5):
                step(b=b)
            n_s = 30 if b <= 64 else 15
            el, _ = timed(n_s, b=b)
            sweep[str(b)] = round(b * n_s / el, 1)
        sweep[str(B)] = round(B * args.steps / elapsed, 1)

    # ---- the other BASELINE configs (configs[2..4]) under the same clock: short child runs of this same script, one at a
    # time, after the headline's timed region (this process only waits; it keeps its 33 GB of HBM, the children fit beside it)
    secondary = None
    if rank == 0 and world == 1 and args.model == MODEL and args.dtype == "bf16" and not args.no_secondary:
        secondary = run_secondary()

    if rank == 0:
        imgs = B * world * args.steps
        value = imgs / elapsed
        peak_chip = (2 * PEAK_BF16_TFLOPS if args.dtype == "fp8" else PEAK_BF16_TFLOPS)
        line = {
            "metric": f"train images/sec (whole node), {args.model} {'256^2 latent' if args.latent == 32 else '4x128x128 latents'}", "value": round(value, 1),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "host_enqueue_ms": host_enqueue_ms,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.model} ({DESC[args.model]}), 4x{S}x{S} synthetic latents + pooled-text cond 1280, "
                                   f"eps-MSE, AdamW lr1e-6 wd0.01 cosine; random-init weights",
                       "per_gpu_batch": B, "global_batch": B * world, "tokens_per_image": None if unet else 256,
                       "parallelism": f"dp{world}"},
            "model_tflops": round(value * step_gflop / 1e3, 1),
            "mfma_frac_whole_step": round(value * step_gflop / 1e3 / (peak_chip * world), 4),
            "final_loss": round(final_loss, 5),
            "roofline": roof,
            "batch_sweep": sweep,
            "comm": comm,
            "secondary": secondary,
        }
        if not args.no_cpu_baseline and world == 1 and args.model == MODEL:
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


SECONDARY = [  # BASELINE.json configs[2..4] at one GPU's share of the work (per-GPU batch as in profiles/)
    ("DiT-B/2 bf16 (config 3)", ["--model", "DiT-B/2", "--batch", "256"]),
    ("DiT-XL/2 fp8 (config 5)", ["--model", "DiT-XL/2", "--dtype", "fp8", "--batch", "192"]),
    ("SDXL-UNet 4x128x128 bf16 (config 4)", ["--model", "SDXL-UNet", "--latent", "128", "--batch", "12"]),
]


def run_secondary(steps=5, warmup=2):
    out = {}
    for name, extra in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-sweep", "--no-secondary", *extra]
        print(f"[bench] secondary: {name} ...", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=420)
            line = json.loads(r.stdout.strip().splitlines()[-1])
            roof = line.get("roofline") or {}
            out[name] = {"images_per_s": line["value"], "ms_per_step": line["ms_per_step"], "per_gpu_batch": line["config"]["per_gpu_batch"],
                         "dtype": line["dtype"], "steps": steps, "warmup": warmup, "model_tflops": line["model_tflops"],
                         "mfma_frac_whole_step": line["mfma_frac_whole_step"], "gemm_family_frac": roof.get("frac"),
                         "final_loss": line["final_loss"], "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:  # a failed child must not take the headline line down with it
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


GEMM_SOURCES = ("gemm_shared.h", "gemm.hip", "gemm_p8.hip", "gemm_p8n.hip", "gemm_p8f.hip")  # (tools/summarize_pmc.py hashes the same list)


def src_sha16():
    import hashlib

    h = hashlib.sha256()
    for name in GEMM_SOURCES:
        with open(os.path.join(ROOT, "uwudiff_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py — OMGSR single-mid-timestep SR throughput on MI355X (see BASELINE.json / BASELINE.md).

    python bench.py --gpus N --steps K --warmup W [--workload s1024|s512|f1024] [--batch B] [--weight-dtype fp32|fp16|bf16]

One process per GPU. With the torchrun env (RANK / WORLD_SIZE ...) this process IS a rank; without it and N > 1 the
parent launches `python -m torch.distributed.run --nproc-per-node N ...` on itself as a child BEFORE anything touches the
GPU and relays its output (never re-execs a GPU process). A "step" is one call of the reference's timed region
(`OMGSR_S_Infer.forward`, infer/omgsr_s_infer_model.py:171-183: NCHW image in the weight dtype already resident in HBM ->
VAE encode -> UNet/Flux at t* -> latent step -> VAE decode -> clamp -> NCHW image) over one batch of synthetic LQ images.
Rank 0 prints ONE JSON line; `value` is whole-job images/s (all ranks' images / max-over-ranks time).

Tiers (`--weight-dtype`, the reference's `--weight_dtype`): fp32 = the accurate tier (fp32 tensors between GEMMs, fp16 MFMA
operands with fp32 accumulation, two-term split operands on the layers omgsr_amd/precision.py names) — the tier that meets the
north-star tolerance (rel-L2 <= 1e-3, PSNR >= 60 dB vs the fp32 reference path) and therefore the default; bf16 (the
reference's default dtype) and fp16 are the fast tiers, reported in `fast_tiers` with their own parity.

Extra legs (rank 0, N == 1):
  roofline      one additional, untimed step with per-launch HIP events on the launch stream (omgsr_timing_*): per kernel
                (igemm_halo_kernel / igemm_dma_kernel / igemm_kernel / attn_kernel) algorithmic FLOPs / summed duration,
                the §8(d) algorithmic figure of the whole workload, and a measured MFMA micro-benchmark peak
  cpu_baseline  the fp32 CPU oracle (oracle/, "port") on ONE image of the same workload; also yields PSNR / rel-L2 of the
                HIP output vs the oracle's output for that image
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import platform
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_DENSE_TFLOPS = 2500.0   # MI355X dense 16-bit MFMA (MI355X_MICROARCH.md: 2.5 PF spec, 2495 TF measured by its micro-benchmark)

WORKLOADS = {
    # name: (family, image side, default batch, latent tile, overlap, algorithmic TFLOP per image [BASELINE.md §3 / SURVEY §8(d)])
    "s512": ("S", 512, 8, 64, 32, 4.436),          # BASELINE.json configs[1]
    "s1024": ("S", 1024, 4, 64, 32, 22.59),        # configs[2]: 1k output, tiled VAE (encoder tile 256, decoder tile 64)
    "f1024": ("F", 1024, 8, 128, 64, 89.8),        # configs[3] (and the per-GPU share of configs[4]: 64 images over 8 GPUs)
}
DTYPES = {"fp32": "float32", "fp16": "float16", "bf16": "bfloat16"}
IGEMM_VARIANTS = {1: "igemm_kernel", 2: "igemm_dma_kernel", 3: "igemm_halo_kernel", 4: "igemm_dma_kernel(split-K)+splitk_reduce_kernel", 5: "igemm_p8_kernel", 6: "igemm_halo_kernel<TAPS=4>", 7: "igemm_halo_multi_kernel", 8: "igemm_halo_multi_kernel<TAPS=4>", 9: "igemm_gmx_kernel",
                  10: "igemm_halo_kernel<GN>", 11: "igemm_halo_multi_kernel<GN>", 12: "igemm_halo_multi_kernel(split-K)+splitk_reduce_kernel",
                  13: "igemm_halo_kernel<fp6>", 14: "igemm_halo_multi_kernel<fp6>", 15: "igemm_halo_multi_kernel<fp6>(split-K)+splitk_reduce_kernel",
                  16: "igemm_halo_kernel<TAPS=4>", 17: "igemm_halo_multi_kernel<TAPS=4>"}       # 16 / 17: the phase form with fp6 correction chunks (named by its form, like rocprofv3's rows)
FP6_VARIANTS = (13, 14, 15, 16, 17)       # correction chunks as fp6: 8 MFMA passes per 64 channels where fp8 takes 16 - 3/4 of the fp8 form's matrix-pipe time


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    # default = the first "1k out" configuration of BASELINE.json's metric that fits one GPU (configs[2])
    ap.add_argument("--workload", default=os.environ.get("OMGSR_BENCH_WORKLOAD", "s1024"), choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (0 = workload default)")
    ap.add_argument("--weight-dtype", default=os.environ.get("OMGSR_BENCH_DTYPE", "fp32"), choices=sorted(DTYPES),
                    help="tier: fp32 = accurate (meets the north-star tolerance; default), bf16 / fp16 = fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fast-tiers", action="store_true", help="skip the short bf16 / fp16 legs")
    ap.add_argument("--no-tiled-vae", action="store_true", help="s1024 only: run the VAE untiled (the reference's shipped default)")
    ap.add_argument("--no-f1024", action="store_true", help="default (s1024, 1 GPU) run only: skip the OMGSR-F 256->1024 record (`workloads.f1024`)")
    ap.add_argument("--no-latency", action="store_true", help="default (s1024, 1 GPU) run only: skip the batch-1 latency record (`workloads.latency_ms_b1`: eager vs hipGraph replay)")
    ap.add_argument("--traffic-file", default="", help="rocprofv3 PMC summary (tools/profile_summary.py) to take `roofline.traffic` from; "
                                                       "default: the newest profiles/r*_traffic.json whose args match this run")
    ap.add_argument("--cpu-runs", type=int, default=1, help="timed runs of the CPU oracle (median reported) after one warm-up on a 512^2 crop")
    ap.add_argument("--all-tiers", action="store_true", help="also time the fp16 tier and the accurate tier with fp8 correction segments (detail file only)")
    ap.add_argument("--detail-file", default=os.environ.get("OMGSR_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="where the full record goes (per-kernel tables, every tier, f1024 / latency records); the stdout line stays compact")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="exercise the N-rank launch / RCCL-shaped broadcast / reporting path on CPU (gloo, reduced models, no kernels)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` makes N ranks itself

def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_spawn_ranks(args) -> int | None:
    """When asked for N > 1 GPUs outside a torchrun environment: run N ranks as CHILD processes (fresh interpreters, spawned
    before this process imports torch.cuda or the HIP library) and return the child's exit code; None = this process is a rank."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------------------------

def build_s(device, rank, world, wdtype, reduced=False):
    import torch
    from omgsr_amd import dist as D
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import seeded_init_
    vcfg = dict(block_out_channels=[32, 64, 128, 128], layers_per_block=1) if reduced else {}
    ucfg = dict(block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4], cross_attention_dim=128) if reduced else {}
    if rank == 0 or world == 1:
        # full-mantissa fp32 values (nothing pre-rounded to 16 bits): what a checkpoint holds after the reference's fp32 LoRA merge
        vae, unet = seeded_init_(AutoencoderKL(**vcfg), 101, rounded=False), seeded_init_(UNet2DConditionModel(**ucfg), 202, rounded=False)
    else:   # peers allocate uninitialised memory and receive rank 0's weights over RCCL
        with torch.device("meta"):
            vae, unet = AutoencoderKL(**vcfg), UNet2DConditionModel(**ucfg)
        vae, unet = vae.to_empty(device=device).to(wdtype), unet.to_empty(device=device).to(wdtype)
    if reduced:     # dry run: no HIP library on this host; the pipeline object is not needed to broadcast / checksum
        pipe = type("DryPipe", (), {})()
        pipe.vae, pipe.unet = vae.to(device=device, dtype=wdtype), unet.to(device=device, dtype=wdtype)
    else:
        pipe = OMGSR_S_Infer(None, None, 273, device, wdtype, vae=vae, unet=unet)
    moved = D.broadcast_module_(pipe.vae) + D.broadcast_module_(pipe.unet)   # RCCL over xGMI (no-op at N=1)
    if not (D.replicas_identical(pipe.vae) and D.replicas_identical(pipe.unet)):
        raise RuntimeError("weight replicas differ after broadcast")
    return pipe, moved


def build_f(device, rank, world, wdtype):
    """OMGSR-F: FLUX.1-dev-shaped DiT (11.9 B params) + FLUX VAE, seeded random weights generated on the GPU."""
    import torch
    from omgsr_amd import dist as D
    from omgsr_amd.diffusers_api import AutoencoderKL, FLUX_VAE_CONFIG, FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer
    from omgsr_amd.testing import seeded_init_, seeded_init_device_
    with torch.device("meta"):
        flux = FluxTransformer2DModel()
    flux = flux.to_empty(device=device).to(wdtype)
    if rank == 0 or world == 1:
        vae = seeded_init_(AutoencoderKL(**FLUX_VAE_CONFIG), 303, rounded=False)
        seeded_init_device_(flux, 404)
    else:
        with torch.device("meta"):
            vae = AutoencoderKL(**FLUX_VAE_CONFIG)
        vae = vae.to_empty(device=device).to(wdtype)
    pipe = OMGSR_F_Infer(None, None, device, wdtype, 244, 1.0, vae=vae, flux_transformer=flux)
    moved = D.broadcast_module_(pipe.vae) + D.broadcast_module_(pipe.flux_transformer)
    if not (D.replicas_identical(pipe.vae) and D.replicas_identical(pipe.flux_transformer)):
        raise RuntimeError("weight replicas differ after broadcast")
    return pipe, moved


def collect_timing(lib_mod):
    from omgsr_amd._lib import TimingEntry
    lib = lib_mod.load()
    n = lib.omgsr_timing_collect(None, 0)
    buf = (TimingEntry * max(n, 1))()
    n = lib.omgsr_timing_collect(buf, n)
    kinds, kernels, shapes, stages = {}, {}, {}, {}
    for i in range(n):
        e = buf[i]
        # MFMA work actually ISSUED, in fp16-equivalent matrix-pipe time: every K segment of the contraction as the
        # kernel runs it (m x n x k with k = R S Cin_slots: a two-term split operand / weight adds a segment each, the mixed-precision form
        # runs 2x the logical channels - its fp8 K-steps cover 64 channels in the time an fp16 step covers 32), the phase-decomposed
        # upsampling convs 4 of their 9 taps. `flops` (omgsr_timing: work_of) is the work HANDED to the kernel: logical channels, 9 taps.
        issued = e.flops
        if e.kind == 1 and e.m > 0 and e.n > 0 and e.k > 0:
            issued = 2.0 * e.m * e.n * e.k * (4.0 / 9.0 if int(e.variant) in (6, 8, 16, 17) else 1.0) * (0.75 if int(e.variant) in FP6_VARIANTS else 1.0)
        k = kinds.setdefault(int(e.kind), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, issued=0.0))
        k["launches"] += 1; k["ms"] += e.ms; k["flops"] += e.flops; k["bytes"] += e.bytes; k["issued"] += issued
        sg = stages.setdefault(int(e.stage), dict(launches=0, ms=0.0, mfma_ms=0.0, mfma_flops=0.0))
        sg["launches"] += 1; sg["ms"] += e.ms
        if e.kind in (1, 2):
            sg["mfma_ms"] += e.ms; sg["mfma_flops"] += e.flops
        if e.kind in (1, 2):
            name = IGEMM_VARIANTS.get(int(e.variant), "igemm?") if e.kind == 1 else "attn_kernel"
            kk = kernels.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, issued=0.0))
            kk["launches"] += 1; kk["ms"] += e.ms; kk["flops"] += e.flops; kk["bytes"] += e.bytes; kk["issued"] += issued
            s = shapes.setdefault((name, int(e.m), int(e.n), int(e.k)), dict(launches=0, ms=0.0, flops=0.0))
            s["launches"] += 1; s["ms"] += e.ms; s["flops"] += e.flops
    table = os.environ.get("OMGSR_KERNEL_TABLE")
    if table:   # per-shape breakdown for DESIGN.md / profiles/
        with open(table, "w") as f:
            f.write("| kernel | M | N | K | launches | total ms | TFLOP/s |\n|---|---|---|---|---|---|---|\n")
            for (name, m, nn, kk), s in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"]):
                tf = s["flops"] / (s["ms"] * 1e-3) / 1e12 if s["ms"] > 0 else 0.0
                f.write(f"| {name} | {m} | {nn} | {kk} | {s['launches']} | {s['ms']:.3f} | {tf:.1f} |\n")
    return kinds, kernels, stages


def cpu_info() -> dict:
    model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count()}


def dry_run_cpu(args) -> None:
    """CPU / gloo rehearsal of the N-rank path: rendezvous, per-rank shard, weight broadcast in buckets, checksum all-reduce,
    barrier-bracketed timing, max-over-ranks, rank-0 JSON. No kernels run (there is no CPU fallback of the product path)."""
    import torch
    from omgsr_amd import dist as D
    rank, _, world = D.init(backend="gloo")
    pipe, moved = build_s(torch.device("cpu"), rank, world, torch.float32, reduced=True)
    B = args.batch or 2
    lo, hi = D.shard_range(B * world, rank, world)
    D.barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (hi - lo))          # stand-in for the step
    D.barrier()
    mine = time.perf_counter() - t1
    elapsed = D.max_over_ranks(mine, torch.device("cpu"))
    fastest = D.min_over_ranks(mine, torch.device("cpu"))
    seen = D.world_size_seen()
    if rank == 0:
        print(json.dumps({"metric": "SR images/sec", "value": round(B * world * args.steps / elapsed, 3), "unit": "images/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "dry_run": True, "config": {"workload": "dry run (CPU, gloo, reduced models, no kernels)", "global_batch": B * world,
                                                      "images_rank0": [lo, hi],
                                                      "parallelism": f"dp{world} (images sharded, weight broadcast {moved / 2**20:.2f} MiB over {'gloo' if world > 1 else 'nothing'})",
                                                      "broadcast_bytes": moved, "world_size": world},
                          "per_rank_ms_per_step": {"min": round(fastest / args.steps * 1e3, 3), "max": round(elapsed / args.steps * 1e3, 3)},
                          "n_ranks_seen": seen, "broadcast_bytes": moved}), flush=True)
    D.shutdown()


def main():
    args = parse()
    rc = maybe_spawn_ranks(args)
    if rc is not None:
        raise SystemExit(rc)
    if args.dry_run_cpu:
        return dry_run_cpu(args)

    import torch
    from omgsr_amd import _lib, dist as D, ops
    from omgsr_amd.testing import synthetic_lq
    wdtype = getattr(torch, DTYPES[args.weight_dtype])

    rank, local_rank, world = D.init()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    _lib.check(_lib.load().omgsr_check_device(), "omgsr_check_device")

    family, side, dbatch, tile, overlap, tflop_per_img = WORKLOADS[args.workload]
    B = args.batch or dbatch
    tiled_vae = args.workload == "s1024" and not args.no_tiled_vae
    t0 = time.time()
    pipe, moved = (build_s if family == "S" else build_f)(device, rank, world, wdtype)
    if tiled_vae:
        pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
    build_secs = time.time() - t0

    # synthetic inputs, resident in HBM in the weight dtype before the timed region (infer/infer_omgsr_s.py:92); per-rank seed
    inp = make_inputs(family, side, B, tile, rank, device, wdtype)
    pipe.vae.posterior_noise = inp["eps"].to(device)
    step = make_step(pipe, family, inp, tile, overlap)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        D.barrier()
        elapsed = time.perf_counter() - t1
    # the contract's number is the MAX over ranks; the fastest rank's time beside it shows stragglers in a SCALE run
    fastest = D.min_over_ranks(elapsed, device)
    elapsed = D.max_over_ranks(elapsed, device)
    ranks_seen = D.world_size_seen()            # a collective: every rank takes part
    images = B * world * args.steps
    value = images / elapsed

    # Everything below is reporting. The timed region above is final; each extra leg runs under `guarded`, which records an error
    # string instead of losing the line, and the stdout line is COMPACT (compact_line: <= 6 KB) - the full record
    # (per-kernel tables, every tier, the f1024 / latency records) goes to --detail-file.
    errors = {}

    def guarded(name, fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except Exception as exc:                       # noqa: BLE001 - a failing extra leg must not take the headline with it
            errors[name] = f"{type(exc).__name__}: {exc}"[:300]
            try:
                ops.set_compute_dtype(wdtype)
                torch.cuda.empty_cache()
            except Exception:                           # noqa: BLE001
                pass
            return None

    roofline, extra = None, {}
    if rank == 0 and not args.no_roofline:
        r = guarded("roofline", roofline_leg, _lib, step, args, tflop_per_img, B, world, elapsed, tiled_vae, wdtype)
        if r is not None:
            roofline, extra = r

    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and family == "S":
        oracle_img = None
        if not args.no_cpu_baseline:
            r = guarded("cpu_baseline", cpu_leg, inp, out[:1], tile, overlap, side, tiled_vae, runs=args.cpu_runs)
            if r is not None:
                cpu_baseline, parity, oracle_img = r
        if not args.no_fast_tiers:
            others = {}
            tiers = ("fp32", "fp16", "bf16") if args.all_tiers else ("fp32", "bf16")
            for name in tiers:
                if name != args.weight_dtype:
                    others[name] = guarded(f"tier_{name}", tier_leg, getattr(torch, DTYPES[name]), device, family, side, B, tile, overlap, tiled_vae, inp, oracle_img)
            # the tier a checkpoint with out-of-fp16-range activations would actually run (precision.RangeFallback, forced)
            others["fp32_range_fallback"] = guarded("tier_fp32_range_fallback", tier_leg, torch.float32, device, family, side, B, tile, overlap, tiled_vae, inp, oracle_img, range_fallback=True)
            if args.all_tiers:
                # the accurate tier with every correction segment as fp8 (round 4's form, OMGSR_MX=8): the same-box reference for what the fp6 segments buy
                others["fp32_mx_fp8_corrections"] = guarded("tier_fp32_mx_fp8", tier_leg, torch.float32, device, family, side, B, tile, overlap, tiled_vae, inp, oracle_img, mx_env="8")
            extra["other_tiers"] = {k: v for k, v in others.items() if v is not None}
            ops.set_compute_dtype(wdtype)

    if rank == 0 and world == 1 and family == "F" and not args.no_cpu_baseline:
        r = guarded("cpu_baseline", cpu_leg_f, device, [args.weight_dtype], side, tile, overlap)
        if r is not None:
            cpu_baseline, parity = r[0], r[1][args.weight_dtype]

    workloads = None
    if rank == 0 and world == 1 and args.workload == "s1024" and not (args.no_f1024 and args.no_latency):
        del pipe, step, out
        torch.cuda.empty_cache()
        workloads = {}
        if not args.no_f1024:
            workloads["f1024"] = guarded("f1024", f1024_record, args, device, _lib)
            ops.set_compute_dtype(wdtype)
        if not args.no_latency:
            workloads["latency_ms_b1"] = guarded("latency_b1", latency_b1_record, args, device, _lib)
            ops.set_compute_dtype(wdtype)
        workloads = {k: v for k, v in workloads.items() if v is not None}

    if rank == 0:
        tier = {"fp32": "accurate tier: fp32 tensors between GEMMs, fp16 MFMA operands (two-term split on the layers of omgsr_amd/precision.py), fp32 accumulation",
                "fp16": "fast tier fp16", "bf16": "fast tier bf16 (the reference's default --weight_dtype)"}[args.weight_dtype]
        full = {
            "metric": "SR images/sec", "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp16" if args.weight_dtype == "fp32" else args.weight_dtype, "data": "synthetic",
            "config": {"workload": f"OMGSR-{family} {side // 4}->{side}, batch={B}/GPU, --weight_dtype {args.weight_dtype} ({tier}), seeded random weights at "
                                   f"{'SD2.1-base' if family == 'S' else 'FLUX.1-dev'} shapes" + (", tiled VAE (VAEHook enc 256 / dec 64)" if tiled_vae else ""),
                       "global_batch": B * world, "latent_tile": tile, "tile_overlap": overlap, "mid_timestep": 273 if family == "S" else 244,
                       "timed_region": "pipe.forward: NCHW image in HBM -> NCHW image (clamped), the reference's own region",
                       "parallelism": f"dp{world} (images sharded, RCCL weight broadcast {moved / 2**20:.1f} MiB)", "world_size": world},
            "args": {"workload": args.workload, "weight_dtype": args.weight_dtype, "batch": B},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
            "per_rank_ms_per_step": {"min": round(fastest / args.steps * 1e3, 3), "max": round(elapsed / args.steps * 1e3, 3)},
            "n_ranks_seen": ranks_seen, "broadcast_bytes": int(moved),
            "setup_s": round(build_secs, 1), **extra,
        }
        if workloads:
            full["workloads"] = workloads
        if errors:
            full["errors"] = errors
        detail_path = None
        try:
            with open(args.detail_file, "w") as f:
                json.dump(full, f, indent=1)
            detail_path = os.path.relpath(args.detail_file, ROOT)
        except OSError as exc:
            errors["detail_file"] = str(exc)[:200]
        print(json.dumps(compact_line(full, family, side, B, tiled_vae, args.weight_dtype, detail_path, errors)), flush=True)
    D.shutdown()


MAX_LINE_BYTES = 6000


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full, family, side, B, tiled_vae, weight_dtype, detail_path, errors):
    """The ONE stdout line (<= MAX_LINE_BYTES): the contract's keys, `roofline` = the DOMINANT kernel's row (work handed per launch /
    its average HIP-event duration) with the igemm family's SURVEY 8(d) figure beside it, `cpu_baseline`, `parity`, the stage split and
    one number per extra leg. Everything else is in the detail file."""
    tier = {"fp32": "accurate tier (fp32 stream, fp16 MFMA operands + split corrections)", "fp16": "fast tier fp16", "bf16": "fast tier bf16"}[weight_dtype]
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = full["config"]
    wl = f"OMGSR-{family} {side // 4}->{side}, batch={B}/GPU, {tier}" + (", tiled VAE 256/64" if tiled_vae else "")
    line["config"] = {"workload": wl[:120], **_pick(cfg, "global_batch", "latent_tile", "tile_overlap", "mid_timestep", "world_size"),
                      "parallelism": f"dp{cfg['world_size']}", "weights": "seeded random, " + ("SD2.1-base" if family == "S" else "FLUX.1-dev") + " shapes",
                      "timed_region": "pipe.forward (reference's region)"}
    rf = full.get("roofline")
    if rf:
        kernels = rf.get("kernels") or {}
        dom = next(iter(kernels), None)
        row = kernels.get(dom, {})
        line["roofline"] = {
            "bound": "mfma", "kernel": dom, "achieved": row.get("achieved_tflops"), "peak": rf["peak"], "unit": "TFLOP/s", "frac": row.get("frac"),
            "basis": "FLOPs handed to the dominant kernel per launch (logical channels, split segments once) / its mean HIP-event duration",
            "launches": row.get("launches"), "kernel_ms": row.get("total_ms"), "avg_us": row.get("avg_us"),
            "issued_mfma_tflops": row.get("issued_tflops"), "peak_measured": rf.get("peak_measured_mfma_microbench"),
            "traffic": row.get("traffic_bytes_per_launch"), "algorithmic_bytes": row.get("bytes_per_launch"),
            "traffic_over_algorithmic": row.get("traffic_over_algorithmic"),
            "traffic_measured_in_this_run": False if rf.get("traffic_source") else None,
            "traffic_source": (rf.get("traffic_source") or "").split(" ")[0] or None, "traffic_kernels_current": rf.get("traffic_kernels_current"),
            "family": {"name": "igemm (all MFMA GEMM / conv kernels)", "basis": "SURVEY 8(d) algorithmic FLOPs / summed kernel time",
                       **_pick(rf, "achieved", "frac", "launches", "kernel_ms", "algorithmic_tflop", "handed_tflop", "issued_mfma_tflop",
                               "traffic", "traffic_over_algorithmic", "pipeline_frac_of_mfma_peak")},
        }
    else:
        line["roofline"] = None
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = ({**_pick(cb, "value", "unit", "cores", "kind"), "sample": str(cb.get("sample", ""))[:160],
                             **_pick(cb, "cpu_model", "nproc")} if cb else None)
    line["parity"] = _pick(full.get("parity"), "rel_l2", "psnr_db", "meets_north_star") or None
    if full.get("stages"):
        line["stages"] = _pick(full["stages"], "encode_ms", "denoiser_ms", "decode_ms", "denoiser_frac_of_mfma_peak", "denoiser_mfma_kernels_frac_of_peak")
    if full.get("kernel_ms_by_family"):
        line["kernel_ms_by_family"] = full["kernel_ms_by_family"]
    also = {}
    for name, leg in (full.get("other_tiers") or {}).items():
        also[f"{full['args']['workload']}_{name}"] = _pick(leg, "ms_per_step", "images_per_s", "rel_l2")
    wls = full.get("workloads") or {}
    f10 = wls.get("f1024") or {}
    for key in ("fp32", "fp32_range_fallback", "bf16"):
        leg = f10.get(key)
        if leg:
            rec = _pick(leg, "ms_per_step", "images_per_s", "batch")
            rec.update(_pick(leg.get("stages"), "denoiser_frac_of_mfma_peak"))
            rec.update(_pick(leg.get("parity"), "rel_l2"))
            also[f"f1024_{key}"] = rec
    lat = wls.get("latency_ms_b1") or {}
    for key in ("fp32", "bf16"):
        if lat.get(key):
            also[f"s512_b1_{key}"] = _pick(lat[key], "eager_ms", "graph_ms", "graph_equals_eager_bitwise")
    if also:
        line["also"] = also
    line.update(_pick(full, "per_rank_ms_per_step", "n_ranks_seen", "broadcast_bytes", "setup_s"))
    if errors:
        line["errors"] = {k: v[:160] for k, v in errors.items()}
    line["detail"] = detail_path
    # the driver keeps an 8 KB tail of stdout: never let the line outgrow it
    for drop in ("also", "kernel_ms_by_family", "stages"):
        if len(json.dumps(line)) <= MAX_LINE_BYTES:
            break
        line.pop(drop, None)
    return line


def make_inputs(family, side, B, tile, rank, device, wdtype):
    import torch
    from omgsr_amd.testing import synthetic_lq
    g = torch.Generator().manual_seed(4321)
    lq_cpu = synthetic_lq(B, side, side, seed=1234 + rank)
    lat_c = 4 if family == "S" else 16
    inp = {"lq_cpu": lq_cpu, "lq": lq_cpu.to(device=device, dtype=wdtype),
           "eps": torch.randn(B, lat_c, side // 8, side // 8, generator=torch.Generator().manual_seed(99 + rank))}
    if family == "S":
        inp["prompt_cpu"] = torch.randn(1, 77, 1024, generator=g)
        inp["prompt"] = inp["prompt_cpu"].to(device=device, dtype=wdtype)
    else:
        from omgsr_amd.pipelines.omgsr_f import prepare_latent_image_ids
        inp["prompt"] = torch.randn(1, 512, 4096, generator=g).to(device=device, dtype=wdtype)
        inp["pooled"] = torch.randn(1, 768, generator=g).to(device=device, dtype=wdtype)
        inp["text_ids"] = torch.zeros(512, 3, device=device, dtype=wdtype)
        inp["image_ids"] = prepare_latent_image_ids(tile // 2, tile // 2, device, wdtype)
    return inp


def make_step(pipe, family, inp, tile, overlap):
    if family == "S":
        return lambda: pipe(inp["lq"], inp["prompt"], tile, overlap)[0]
    return lambda: pipe(inp["lq"], inp["prompt"], inp["pooled"], inp["text_ids"], inp["image_ids"], tile, overlap)[0]


DENOISER_TFLOP = {"s512": 0.804, "s1024": 9 * 0.804, "f1024": 74.4}      # per image (0.804 per 64 x 64 UNet tile, 74.4 per Flux forward)


def find_traffic(args, workload, weight_dtype, B):
    """`roofline.traffic` comes from rocprofv3 PMC passes, which cannot run inside this process: tools/profile_round.sh collects them
    for one (workload, tier, batch) and tools/profile_summary.py writes profiles/r<round>_traffic.json. Take --traffic-file, else
    the newest such file whose recorded args equal this run's; a stale / absent file gives traffic = null, never a wrong number.
    Returns (family bytes per launch, per-kernel rows {bench kernel name: row}, path, kernel-source digest)."""
    import glob
    cands = [args.traffic_file] if args.traffic_file else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True)
    for tpath in cands:
        try:
            tj = json.load(open(tpath))
        except (OSError, ValueError):
            continue
        if tj.get("workload") == workload and tj.get("weight_dtype") == weight_dtype and tj.get("batch") == B:
            fam = (tj.get("families") or {}).get("igemm")
            if fam:
                return fam, tj.get("kernels") or {}, os.path.relpath(tpath, ROOT), tj.get("source_digest")
    return None, None, None, None


def attach_traffic(roofline, per_kernel, fam, krows):
    """Per-kernel measured HBM bytes next to the algorithmic `bytes_per_launch` - only where the PMC run saw EXACTLY the launches this
    process timed for that kernel (a mean over a subset of the launches is not a traffic figure). rocprofv3 sees
    kernel names, bench.py dispatch variants: the split-K variant is an igemm_dma_kernel launch plus a splitk_reduce_kernel launch."""
    mine = {k: v["launches"] for k, v in per_kernel.items()}
    sk = mine.get("igemm_dma_kernel(split-K)+splitk_reduce_kernel", 0)
    expect = {k: v for k, v in mine.items() if "(split-K)" not in k}
    hsk = mine.get("igemm_halo_multi_kernel(split-K)+splitk_reduce_kernel", 0)      # (round 5: chunk ranges of the halo-tile kernel as one launch group)
    if sk:
        expect["igemm_dma_kernel"] = expect.get("igemm_dma_kernel", 0) + sk
    if hsk:
        expect["igemm_halo_multi_kernel"] = expect.get("igemm_halo_multi_kernel", 0) + hsk
    hsk6 = mine.get("igemm_halo_multi_kernel<fp6>(split-K)+splitk_reduce_kernel", 0)
    if hsk6:
        expect["igemm_halo_multi_kernel<fp6>"] = expect.get("igemm_halo_multi_kernel<fp6>", 0) + hsk6
    if sk or hsk or hsk6:
        expect["splitk_reduce_kernel"] = sk + hsk + hsk6
    ok_all, total_meas, total_n = True, 0.0, 0
    for name, n in expect.items():
        row = krows.get(name)
        seen = row.get("launches_per_pipeline_pass") if row else None
        ok = row is not None and seen is not None and abs(seen - n) < 0.5
        ok_all = ok_all and (ok or name == "attn_kernel")
        if name.startswith(("igemm", "splitk")) and ok:
            total_meas += row["hbm_bytes_per_launch"] * n
            total_n += n
        tgt = per_kernel.get(name)
        if tgt is None and name == "splitk_reduce_kernel":
            continue
        if tgt is not None:
            tgt["traffic_bytes_per_launch"] = round(row["hbm_bytes_per_launch"]) if ok else None
            tgt["traffic_launches_seen"] = seen
            if ok and tgt["bytes_per_launch"] > 0 and "(split-K)" not in name and not (name == "igemm_dma_kernel" and sk) and not (name == "igemm_halo_multi_kernel" and hsk) and not (name == "igemm_halo_multi_kernel<fp6>" and hsk6):
                tgt["traffic_over_algorithmic"] = round(row["hbm_bytes_per_launch"] / tgt["bytes_per_launch"], 3)
    n_igemm = sum(n for k, n in expect.items() if k.startswith(("igemm", "splitk")))
    if ok_all and total_n == n_igemm and roofline["launches"] > 0:
        # family figure: measured bytes of EVERY igemm launch of the step / the step's igemm launches as bench.py counts them
        roofline["traffic"] = round(total_meas / roofline["launches"])
        roofline["traffic_over_algorithmic"] = round(total_meas / roofline["launches"] / max(roofline["algorithmic_bytes_per_launch"], 1), 3)
    else:
        roofline["traffic"] = None
        roofline["traffic_note"] = "PMC launch counts do not match this step's launches per kernel: no family figure quoted"
    roofline["traffic_launch_counts_match"] = ok_all


def roofline_leg(_lib, step, args, tflop_per_img, B, world, elapsed, tiled_vae, wdtype, workload=None, weight_dtype=None, steps=None):
    import torch
    workload, weight_dtype, steps = workload or args.workload, weight_dtype or args.weight_dtype, steps or args.steps
    lib = _lib.load()
    lib.omgsr_timing_reset(); lib.omgsr_timing_enable(1)
    with torch.no_grad():
        step()
    kinds, kernels, stage_recs = collect_timing(_lib)
    lib.omgsr_timing_enable(0); lib.omgsr_timing_reset()
    peak_meas = C.c_float(0.0)
    _lib.check(lib.omgsr_mfma_peak(4096, C.byref(peak_meas), torch.cuda.current_stream().cuda_stream), "omgsr_mfma_peak")
    per_kernel = {}
    for name, k in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"]):
        # per kernel (SURVEY 8(d) / VERDICT r5 item 1): `achieved_tflops` / `frac` are ALGORITHMIC - the FLOPs the launches were handed
        # (logical channels, a split operand's extra K segments counted once, 9 taps for the phase form, tile overlap included: it is
        # work of that launch) / their summed HIP-event time / the 2.5 PF peak. `issued_tflops` / `issued_frac` is the MFMA work the
        # kernel actually ran for that, in fp16-equivalent matrix-pipe time (what SQ_VALU_MFMA_BUSY_CYCLES reconciles with).
        iss = k["issued"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
        hand = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
        per_kernel[name] = {"launches": k["launches"], "total_ms": round(k["ms"], 3), "avg_us": round(k["ms"] * 1e3 / k["launches"], 2),
                            "handed_tflop": round(k["flops"] / 1e12, 3), "issued_mfma_tflop": round(k["issued"] / 1e12, 3),
                            "achieved_tflops": round(hand, 1), "frac": round(hand / PEAK_DENSE_TFLOPS, 4),
                            "issued_tflops": round(iss, 1), "issued_frac": round(iss / PEAK_DENSE_TFLOPS, 4),
                            "issued_frac_of_measured_peak": round(iss / max(float(peak_meas.value), 1.0), 4),
                            "bytes_per_launch": round(k["bytes"] / k["launches"])}
    roofline = None
    ig, at = kinds.get(1), kinds.get(2)
    if ig and ig["ms"] > 0 and per_kernel:
        dom = next(iter(per_kernel))
        attn_tflop = (at["flops"] / 1e12) if at else 0.0
        # §8(d): the algorithmic figure of the workload (untiled VAE) - the tiled VAE's overlap recompute and the
        # two-term split's duplicated channels are overhead, not useful work. MFMA kernels = igemm family + attention.
        alg_igemm = tflop_per_img * B - attn_tflop
        mfma_ms = ig["ms"] + (at["ms"] if at else 0.0)
        ach = alg_igemm * 1e12 / (ig["ms"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": f"igemm family (dominant: {dom}); per-kernel rows in `kernels`",
                    "achieved": round(ach, 2), "peak": PEAK_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_DENSE_TFLOPS, 4),
                    "peak_measured_mfma_microbench": round(float(peak_meas.value), 1),
                    "frac_of_measured_peak": round(ach / max(float(peak_meas.value), 1.0), 4),
                    "traffic": None,
                    "launches": ig["launches"], "kernel_ms": round(ig["ms"], 3),
                    "algorithmic_tflop": round(alg_igemm, 3), "handed_tflop": round(ig["flops"] / 1e12, 3),
                    "issued_mfma_tflop": round(ig["issued"] / 1e12, 3),
                    "issued_frac_of_peak": round(ig["issued"] / (ig["ms"] * 1e-3) / 1e12 / PEAK_DENSE_TFLOPS, 4),
                    "issued_frac_of_measured_peak": round(ig["issued"] / (ig["ms"] * 1e-3) / 1e12 / max(float(peak_meas.value), 1.0), 4),
                    # the whole step (every kernel, HBM-bound ones included) on the same algorithmic basis, driver-checkable: tflop_per_img x B / ms_per_step
                    "pipeline_frac_of_mfma_peak": round(tflop_per_img * B * steps / elapsed / PEAK_DENSE_TFLOPS, 4) if world == 1 else None,
                    # `frac` divides the §8(d) ALGORITHMIC FLOPs by the kernels' time: the accurate tier's extra K segments (split operands /
                    # weights: up to 3 MFMA passes per layer) and the tiled VAE's overlap recompute are overhead there, not work. The
                    # kernels' own rate on the work they were handed (per-launch FLOPs incl. tile overlap, split segments counted once):
                    "frac_of_handed_work": round(ig["flops"] / (ig["ms"] * 1e-3) / 1e12 / PEAK_DENSE_TFLOPS, 4),
                    "algorithmic_basis": f"{tflop_per_img} TFLOP/image x {B} images (SURVEY 8(d), untiled VAE) minus {attn_tflop:.3f} TFLOP run by attn_kernel",
                    "achieved_all_mfma_kernels": round(tflop_per_img * B / (mfma_ms * 1e-3), 2),
                    "algorithmic_bytes_per_launch": round(ig["bytes"] / ig["launches"]),
                    "kernels": per_kernel}
        fam, krows, tsrc, tdigest = find_traffic(args, workload, weight_dtype, B)
        if fam is not None:
            from omgsr_amd.build import _source_digest
            if krows:
                attach_traffic(roofline, per_kernel, fam, krows)
            else:            # a pre-round-4 file (family mean over whatever its regex matched): not trusted
                roofline["traffic_note"] = f"{tsrc} has no per-kernel rows (pre-round-4 summary): regenerate with tools/profile_round.sh"
            roofline["traffic_source"] = f"{tsrc} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, bytes per launch)"
            roofline["traffic_kernels_current"] = (tdigest == _source_digest()) if tdigest else None     # False: kernels changed since the PMC run
    names = {1: "igemm", 2: "attention", 3: "groupnorm", 4: "layernorm", 5: "elementwise", 6: "softmax"}
    snames = {0: "other", 1: "encode", 2: "denoiser", 3: "decode"}
    stages = {f"{snames.get(k, str(k))}_ms": round(v["ms"], 3) for k, v in sorted(stage_recs.items())}
    den = stage_recs.get(2)
    if den and den["ms"] > 0:
        dt = DENOISER_TFLOP.get(workload, 0.0) * B
        # the north-star's 70 % target is stated on THIS quantity: algorithmic FLOPs of the UNet / DiT forward over the time the
        # denoiser stage takes (all of its kernels, HBM-bound ones included), against the 2.5 PFLOP/s dense 16-bit peak
        stages["denoiser_algorithmic_tflop"] = round(dt, 3)
        stages["denoiser_frac_of_mfma_peak"] = round(dt / (den["ms"] * 1e-3) / PEAK_DENSE_TFLOPS, 4)
        stages["denoiser_mfma_kernels_frac_of_peak"] = round(dt / (den["mfma_ms"] * 1e-3) / PEAK_DENSE_TFLOPS, 4) if den["mfma_ms"] > 0 else None
    extra = {"stages": stages,
             "kernel_ms_by_family": {names.get(k, str(k)): round(v["ms"], 3) for k, v in sorted(kinds.items())},
             # HBM-bound families: algorithmic bytes (each tensor read / written once) over their summed HIP-event time, vs 8 TB/s peak
             "hbm_gbps_by_family": {names.get(k, str(k)): round(v["bytes"] / (v["ms"] * 1e-3) / 1e9) for k, v in sorted(kinds.items())
                                    if k in (3, 4, 5, 6) and v["ms"] > 0},
             "pipeline_frac_of_mfma_peak": round(tflop_per_img * B * steps / elapsed / PEAK_DENSE_TFLOPS, 4) if world == 1 else None}
    return roofline, extra


def cpu_leg(inp, hip_out_nchw, tile, overlap, side, tiled_vae=False, runs=3):
    """fp32 CPU oracle on ONE image of the workload (bounded sample): one untimed warm-up on a 512^2 crop (thread pool, allocator,
    oneDNN primitive caches), then `runs` timed passes - the median is reported - and parity of the HIP output."""
    import torch
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef, TiledVaeRef
    # 16 threads is the fastest eager-fp32 configuration on the GPU box's 2 x EPYC 9575F (measured with
    # tools/cpu_threads_probe.py: 16 thr 1.13 TFLOP/s, 32 thr 0.77, 64 thr 0.52, 128 thr 0.24)
    cores = int(os.environ.get("OMGSR_CPU_THREADS", min(16, os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    vae, unet = seeded_init_(R.AutoencoderKL(), 101, rounded=False).eval(), seeded_init_(R.UNet2DConditionModel(), 202, rounded=False).eval()
    vae.posterior_noise = inp["eps"][:1]
    ref = OmgsrSRef(TiledVaeRef(vae, 256, 64) if tiled_vae else vae, unet, R.DDPMScheduler().alphas_cumprod[273], 273)
    times = []
    with torch.no_grad():
        if runs > 1:
            vae.posterior_noise = inp["eps"][:1, :, :64, :64]
            OmgsrSRef(vae, unet, R.DDPMScheduler().alphas_cumprod[273], 273)(inp["lq_cpu"][:1, :, :512, :512], inp["prompt_cpu"], tile, overlap)
            vae.posterior_noise = inp["eps"][:1]
        for _ in range(max(1, runs)):
            t0 = time.perf_counter()
            img = ref(inp["lq_cpu"][:1], inp["prompt_cpu"], tile, overlap)
            times.append(time.perf_counter() - t0)
    secs = sorted(times)[len(times) // 2]
    got = hip_out_nchw.float().cpu()
    parity = {"vs": "fp32 CPU oracle, same weights/inputs/eps, image 0", "rel_l2": round(rel_l2(got, img), 6),
              "psnr_db": round(psnr(got, img), 2), "north_star": "rel_l2 <= 1e-3 and psnr >= 60 dB",
              "meets_north_star": bool(rel_l2(got, img) <= 1e-3 and psnr(got, img) >= 60.0)}
    base = {"value": round(1.0 / secs, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 image {side // 4}->{side}, fp32 eager PyTorch oracle, median of {len(times)} timed runs after a warm-up "
                      f"({', '.join(f'{t:.1f}' for t in times)} s), torch threads={torch.get_num_threads()}",
            "runs_s": [round(t, 2) for t in times], **cpu_info(), "torch": torch.__version__}
    return base, parity, img


def cpu_leg_f(device, tiers, side, tile, overlap):
    """OMGSR-F parity + CPU baseline on a bounded sample: ONE image through a pipeline whose DiT has the full FLUX.1-dev WIDTH
    (D 3072, 24 x 128 heads, 4096 + 512 tokens) but 2 + 2 of its 19 + 38 blocks (the full-depth fp32 oracle streams 48 GB of
    weights and ~90 TFLOP through the host: 140 s, tests/test_flux_fullsize_gpu.py runs THAT comparison), seeded on the CPU with
    full fp32 mantissas so the oracle holds bit-identical weights; the VAE is the full FLUX VAE. One oracle run, every tier in
    `tiers` compared with it. Returns (cpu_baseline, {tier: parity})."""
    import torch
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, FLUX_VAE_CONFIG, FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer, prepare_latent_image_ids
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrFRef
    from oracle.pipeline_ref import prepare_latent_image_ids as ref_ids
    cores = int(os.environ.get("OMGSR_CPU_THREADS", min(16, os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    cfg = dict(num_layers=2, num_single_layers=2)
    ov = seeded_init_(R.AutoencoderKL(**FLUX_VAE_CONFIG), 303, rounded=False).eval()
    of = seeded_init_(R.FluxTransformer2DModel(**cfg), 404, rounded=False).eval()
    g = torch.Generator().manual_seed(4321)
    x = synthetic_lq(1, side, side, seed=1234)
    eps = torch.randn(1, 16, side // 8, side // 8, generator=torch.Generator().manual_seed(99))
    pe, pooled = torch.randn(1, 512, 4096, generator=g), torch.randn(1, 768, generator=g)
    tids, iids = torch.zeros(512, 3), ref_ids(tile // 2, tile // 2)
    ov.posterior_noise = eps
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = OmgsrFRef(ov, of, 244, 1.0)(x, pe, pooled, tids, iids, tile, overlap)
        secs = time.perf_counter() - t0
    sdv, sdf = ov.state_dict(), of.state_dict()
    del ov, of
    parity = {}
    for tier in tiers:
        wdtype = getattr(torch, DTYPES["fp32" if tier.startswith("fp32") else tier])
        pv, pf = AutoencoderKL(**FLUX_VAE_CONFIG), FluxTransformer2DModel(**cfg)
        pv.load_state_dict(sdv); pf.load_state_dict(sdf)
        pf.round_timestep_to_weight_dtype = False          # the fp32 oracle conditions on the exact sigma(t*)
        pipe = OMGSR_F_Infer(None, None, device, wdtype, 244, 1.0, vae=pv, flux_transformer=pf)
        if tier == "fp32_range_fallback":
            pipe.range_fallback.enter()
        pipe.vae.posterior_noise = eps.to(device)
        with torch.no_grad():
            got, _ = pipe(x.to(device=device, dtype=wdtype), pe.to(device=device, dtype=wdtype), pooled.to(device=device, dtype=wdtype),
                          tids.to(device=device, dtype=wdtype), prepare_latent_image_ids(tile // 2, tile // 2, device, wdtype), tile, overlap)
        got = got.float().cpu()
        e, p = rel_l2(got, ref), psnr(got, ref)
        parity[tier] = {"vs": "fp32 CPU oracle (full-mantissa fp32 weights), same inputs/eps, 1 image, FLUX.1-dev width with 2+2 of 19+38 DiT blocks + "
                              "the full FLUX VAE; the full-depth comparison is tests/test_flux_fullsize_gpu.py",
                        "rel_l2": round(e, 6), "psnr_db": round(p, 2), "north_star": "rel_l2 <= 1e-3 and psnr >= 60 dB",
                        "meets_north_star": bool(e <= 1e-3 and p >= 60.0)}
        if tier == "fp32_range_fallback":
            pipe.range_fallback.reset()
        del pipe, pv, pf
        torch.cuda.empty_cache()
    base = {"value": round(1.0 / secs, 5), "unit": "images/s (reduced-depth DiT: 20.3 of the full pipeline's 89.8 TFLOP)", "cores": cores, "kind": "port",
            "sample": f"1 image {side // 4}->{side}, fp32 eager PyTorch oracle with a 2+2-block DiT, {secs:.1f} s, torch threads={torch.get_num_threads()}",
            **cpu_info(), "torch": torch.__version__}
    return base, parity


def f1024_record(args, device, _lib):
    """BASELINE configs[3] (OMGSR-F 256->1024, FLUX.1-dev-shaped DiT at full depth + FLUX VAE) inside the default run, so that the
    89.8 TFLOP/image configuration is timed under the same driver clock as the headline: the accurate tier and the reference's default bf16, both at
    batch 8 (the per-GPU share of configs[4]), each with its own roofline / stages, plus parity and a
    CPU baseline on the bounded (reduced-depth) sample of cpu_leg_f."""
    import torch
    from omgsr_amd import ops
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer
    family, side, _, tile, overlap, tflop = WORKLOADS["f1024"]
    rec = {"config": "OMGSR-F 256->1024, one FluxTransformer2DModel step at sigma(t*=244), 19+38 blocks at FLUX.1-dev shapes + FLUX VAE (untiled), "
                     "seeded random weights generated on the GPU", "algorithmic_tflop_per_image": tflop}
    t0 = time.time()
    pipe, _ = build_f(device, 0, 1, torch.float32)
    rec["setup_s"] = round(time.time() - t0, 1)
    nsteps = 3
    # "fp32_range_fallback": the accurate tier as a real FLUX checkpoint would run it: FLUX activations leave the fp16 range
    # (the reason the reference defaults to bf16, infer/infer_omgsr_f.py:137), the range guard fires on the first image and the pipeline
    # stays on bf16 operands with every operand and weight split. Forced here (seeded weights never clip) with range_fallback.enter().
    for key, B in (("fp32", 8), ("fp32_range_fallback", 8), ("bf16", 8)):
        tier = "fp32" if key.startswith("fp32") else key
        wd = getattr(torch, DTYPES[tier])
        if key == "fp32_range_fallback":
            pipe.range_fallback.enter()
        if tier != "fp32":      # the same modules, cast in place to the reference's default dtype
            pipe.range_fallback.reset()
            pipe = OMGSR_F_Infer(None, None, device, wd, 244, 1.0, vae=pipe.vae, flux_transformer=pipe.flux_transformer)
        inp = make_inputs(family, side, B, tile, 0, device, wd)
        pipe.vae.posterior_noise = inp["eps"].to(device)
        step = make_step(pipe, family, inp, tile, overlap)
        with torch.no_grad():
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nsteps):
                step()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t1
        roofline, extra = (None, {}) if args.no_roofline else roofline_leg(_lib, step, args, tflop, B, 1, elapsed, False, wd, workload="f1024",
                                                                           weight_dtype=key, steps=nsteps)
        rec[key] = {"images_per_s": round(B * nsteps / elapsed, 3), "ms_per_step": round(elapsed / nsteps * 1e3, 3), "batch": B, "steps": nsteps,
                    "warmup": 1, "roofline": roofline, **extra}
        if key == "fp32_range_fallback":
            rec[key]["mode"] = "fp32 stream, bf16 MFMA operands, every operand and weight split (3 K segments everywhere); sticky"
            rec[key]["sticky"] = bool(pipe.range_fallback.sticky)
        del step, inp
    del pipe
    torch.cuda.empty_cache()
    if not args.no_cpu_baseline:
        rec["cpu_baseline"], par = cpu_leg_f(device, ["fp32", "fp32_range_fallback", "bf16"], side, tile, overlap)
        for tier, pr in par.items():
            rec[tier]["parity"] = pr
    return rec


def latency_b1_record(args, device, _lib):
    """The reference's OWN operating point: batch 1, one 128 -> 512 image per call (infer/infer_omgsr_s.py:92-93),
    as wall-clock latency of pipe.forward - eager (every launch issued from Python) and as a hipGraph replay (pipe.enable_graphs():
    one host call per image) - next to the GPU time of the same step (sum of the kernels' HIP-event times): `gpu_busy_frac` says how
    launch-bound the eager path is. Replay and eager outputs are compared bit for bit."""
    import torch
    from omgsr_amd import ops
    family, side, _, tile, overlap, tflop = WORKLOADS["s512"]
    rec = {"config": "OMGSR-S 128->512, batch 1, untiled VAE, latent 64 x 64 (one UNet tile)", "algorithmic_tflop_per_image": tflop}
    lib = _lib.load()
    for tier in ("fp32", "bf16"):
        wd = getattr(torch, DTYPES[tier])
        pipe, _ = build_s(device, 0, 1, wd)
        inp = make_inputs(family, side, 1, tile, 0, device, wd)
        pipe.vae.posterior_noise = inp["eps"].to(device)
        step = make_step(pipe, family, inp, tile, overlap)

        def timed(n):
            with torch.no_grad():
                torch.cuda.synchronize()
                ts = []
                for _ in range(n):
                    t0 = time.perf_counter()
                    out = step()                      # forward() synchronises on both sides, like the reference's
                    ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2] * 1e3, out
        timed(3)
        eager_ms, out_eager = timed(20)
        lib.omgsr_timing_reset(); lib.omgsr_timing_enable(1)
        with torch.no_grad():
            step()
        kinds, kernels, _ = collect_timing(_lib)
        lib.omgsr_timing_enable(0); lib.omgsr_timing_reset()
        gpu_ms = sum(k["ms"] for k in kinds.values())
        launches = sum(k["launches"] for k in kinds.values())
        pipe.enable_graphs(True)
        timed(3)                                      # first call eager, second captures, third replays
        graph_ms, out_graph = timed(20)
        rec[tier] = {"eager_ms": round(eager_ms, 3), "graph_ms": round(graph_ms, 3), "gpu_kernel_ms": round(gpu_ms, 3), "timed_launches": launches,
                     "gpu_busy_frac_eager": round(gpu_ms / eager_ms, 3), "gpu_busy_frac_graph": round(gpu_ms / graph_ms, 3),
                     "images_per_s_graph": round(1e3 / graph_ms, 2), "graph_replays": pipe.graphs.replays, "graph_captures": pipe.graphs.captures,
                     "graph_equals_eager_bitwise": bool(torch.equal(out_eager, out_graph)),
                     "frac_of_mfma_peak_graph": round(tflop / (graph_ms * 1e-3) / PEAK_DENSE_TFLOPS, 4)}
        del pipe, step, inp
        torch.cuda.empty_cache()
    return rec


def tier_leg(wdtype, device, family, side, B, tile, overlap, tiled_vae, inp, oracle_img, range_fallback=False, mx_env=None):
    """The same workload in another --weight_dtype tier: a short timed run + parity of image 0 against the same oracle output.
    range_fallback (accurate tier only): the mode a checkpoint whose activations leave the fp16 range runs in - forced
    with pipe.range_fallback.enter(): bf16 operands (fp32's exponent range), EVERY operand and weight as a two-term split = three K
    segments on every layer, sticky - instead of waiting for an fp16 operand to clip."""
    import torch
    from omgsr_amd.testing import psnr, rel_l2
    saved = os.environ.get("OMGSR_MX")
    if mx_env is not None:
        os.environ["OMGSR_MX"] = mx_env          # precision.set_mx reads it when the pipeline applies its policy
    try:
        pipe, _ = build_s(device, 0, 1, wdtype)
    finally:
        if mx_env is not None:
            if saved is None:
                os.environ.pop("OMGSR_MX", None)
            else:
                os.environ["OMGSR_MX"] = saved
    if range_fallback:
        pipe.range_fallback.enter()
    if tiled_vae:
        pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
    pipe.vae.posterior_noise = inp["eps"].to(device)
    inp2 = dict(inp, lq=inp["lq_cpu"].to(device=device, dtype=wdtype), prompt=inp["prompt_cpu"].to(device=device, dtype=wdtype))
    step = make_step(pipe, family, inp2, tile, overlap)
    with torch.no_grad():
        out = step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            out = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
    leg = {"images_per_s": round(B / dt, 3), "ms_per_step": round(dt * 1e3, 3), "steps": 3}
    if oracle_img is not None:
        got = out[:1].float().cpu()
        leg.update(rel_l2=round(rel_l2(got, oracle_img), 6), psnr_db=round(psnr(got, oracle_img), 2))
    if mx_env is not None:
        leg["mode"] = f"accurate tier under OMGSR_MX={mx_env}: correction segments as fp8 (e4m3, per-tensor scales) on every mixed-precision layer"
    if range_fallback:
        leg["mode"] = "fp32 stream, bf16 MFMA operands, every operand and weight split (3 K segments everywhere); sticky"
        leg["sticky"] = bool(pipe.range_fallback.sticky)
        pipe.range_fallback.reset()
    del pipe
    torch.cuda.empty_cache()
    return leg


if __name__ == "__main__":
    main()

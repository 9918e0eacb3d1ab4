#!/usr/bin/env python
"""bench.py — OMGSR single-mid-timestep SR throughput on MI355X (see BASELINE.json / BASELINE.md).

    python bench.py --gpus N --steps K --warmup W [--workload s512|s1024|f1024] [--batch B]

One process per GPU (torchrun env for N > 1). A "step" is one pass of the hot path (VAE encode ->
UNet/Flux at t* -> latent step -> VAE decode, the reference's timed region
infer/omgsr_s_infer_model.py:171-183) over one batch of synthetic LQ images already resident in HBM.
Rank 0 prints ONE JSON line; `value` is whole-job images/s (all ranks' images / max-over-ranks time).

Extra legs (rank 0, N == 1):
  roofline      one additional, untimed step with per-launch HIP events on the launch stream
                (omgsr_timing_*): algorithmic FLOPs of every implicit-GEMM launch / their summed duration
  cpu_baseline  the fp32 CPU oracle (oracle/, "port") on ONE image of the same workload, all host cores;
                also yields PSNR / rel-L2 of the HIP output vs the oracle's output for that image
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WDTYPE = torch.bfloat16          # --weight-dtype (the reference's --weight_dtype): bf16 (default) or fp16, same MFMA rate
PEAK_BF16_DENSE_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: 2.5 PF, measured 2495 TF)

WORKLOADS = {
    # name: (family, image side, default batch, latent tile, overlap, algorithmic TFLOP per image [BASELINE.md §3])
    "s512": ("S", 512, 8, 64, 32, 4.436),          # BASELINE.json configs[1]
    "s1024": ("S", 1024, 4, 64, 32, 22.59),        # configs[2]: 1k output, tiled VAE (encoder tile 256, decoder tile 64)
    "f1024": ("F", 1024, 1, 128, 64, 89.8),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    # default = the first "1k out" configuration of BASELINE.json's metric that fits one GPU (configs[2])
    ap.add_argument("--workload", default=os.environ.get("OMGSR_BENCH_WORKLOAD", "s1024"), choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (0 = workload default)")
    ap.add_argument("--weight-dtype", default="bf16", choices=["bf16", "fp16"], help="the kernels' 16-bit element type")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-tiled-vae", action="store_true", help="s1024 only: run the VAE untiled (the reference's shipped default)")
    return ap.parse_args()


def build_s(device, rank, world):
    from omgsr_amd import dist as D
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import seeded_init_
    if rank == 0 or world == 1:
        vae, unet = seeded_init_(AutoencoderKL(), 101), seeded_init_(UNet2DConditionModel(), 202)
    else:   # peers allocate uninitialised HBM and receive rank 0's weights over RCCL
        with torch.device("meta"):
            vae, unet = AutoencoderKL(), UNet2DConditionModel()
        vae, unet = vae.to_empty(device=device).to(WDTYPE), unet.to_empty(device=device).to(WDTYPE)
    pipe = OMGSR_S_Infer(None, None, 273, device, WDTYPE, vae=vae, unet=unet)
    moved = D.broadcast_module_(pipe.vae) + D.broadcast_module_(pipe.unet)   # RCCL over xGMI (no-op at N=1)
    if not (D.replicas_identical(pipe.vae) and D.replicas_identical(pipe.unet)):
        raise RuntimeError("weight replicas differ after broadcast")
    return pipe, moved


def build_f(device, rank, world):
    """OMGSR-F: FLUX.1-dev-shaped DiT (11.9 B params) + FLUX VAE, seeded random weights generated on the GPU."""
    from omgsr_amd import dist as D
    from omgsr_amd.diffusers_api import AutoencoderKL, FLUX_VAE_CONFIG, FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer
    from omgsr_amd.testing import seeded_init_, seeded_init_device_
    with torch.device("meta"):
        flux = FluxTransformer2DModel()
    flux = flux.to_empty(device=device).to(WDTYPE)
    if rank == 0 or world == 1:
        vae = seeded_init_(AutoencoderKL(**FLUX_VAE_CONFIG), 303)
        seeded_init_device_(flux, 404)
    else:
        with torch.device("meta"):
            vae = AutoencoderKL(**FLUX_VAE_CONFIG)
        vae = vae.to_empty(device=device).to(WDTYPE)
    pipe = OMGSR_F_Infer(None, None, device, WDTYPE, 244, 1.0, vae=vae, flux_transformer=flux)
    moved = D.broadcast_module_(pipe.vae) + D.broadcast_module_(pipe.flux_transformer)
    if not (D.replicas_identical(pipe.vae) and D.replicas_identical(pipe.flux_transformer)):
        raise RuntimeError("weight replicas differ after broadcast")
    return pipe, moved


def collect_roofline(lib_mod):
    from omgsr_amd._lib import TimingEntry
    lib = lib_mod.load()
    n = lib.omgsr_timing_collect(None, 0)
    buf = (TimingEntry * max(n, 1))()
    n = lib.omgsr_timing_collect(buf, n)
    kinds, shapes = {}, {}
    for i in range(n):
        e = buf[i]
        k = kinds.setdefault(int(e.kind), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        k["launches"] += 1; k["ms"] += e.ms; k["flops"] += e.flops; k["bytes"] += e.bytes
        if e.kind in (1, 2):
            s = shapes.setdefault((int(e.kind), int(e.m), int(e.n), int(e.k)), dict(launches=0, ms=0.0, flops=0.0))
            s["launches"] += 1; s["ms"] += e.ms; s["flops"] += e.flops
    table = os.environ.get("OMGSR_KERNEL_TABLE")
    if table:   # per-shape breakdown for DESIGN.md / profiles/
        with open(table, "w") as f:
            f.write("| kind | M | N | K | launches | total ms | TFLOP/s |\n|---|---|---|---|---|---|---|\n")
            for (kind, m, nn, kk), s in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"]):
                tf = s["flops"] / (s["ms"] * 1e-3) / 1e12 if s["ms"] > 0 else 0.0
                f.write(f"| {'igemm' if kind == 1 else 'attn'} | {m} | {nn} | {kk} | {s['launches']} | {s['ms']:.3f} | {tf:.1f} |\n")
    return kinds


def main():
    args = parse()
    global WDTYPE
    WDTYPE = torch.bfloat16 if args.weight_dtype == "bf16" else torch.float16
    from omgsr_amd import _lib, dist as D
    from omgsr_amd.testing import psnr, rel_l2, synthetic_lq

    rank, local_rank, world = D.init()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    _lib.check(_lib.load().omgsr_check_device(), "omgsr_check_device")

    family, side, dbatch, tile, overlap, tflop_per_img = WORKLOADS[args.workload]
    B = args.batch or dbatch
    t0 = time.time()
    pipe, moved = (build_s if family == "S" else build_f)(device, rank, world)
    tiled_vae = args.workload == "s1024" and not args.no_tiled_vae
    if tiled_vae:
        pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
    build_s_secs = time.time() - t0

    # synthetic inputs, resident in HBM before the timed region (per-rank seed: every rank has its own images)
    g = torch.Generator().manual_seed(4321)
    lq_cpu = synthetic_lq(B, side, side, seed=1234 + rank)
    lq = ops_nhwc(lq_cpu.to(device))
    lat_c = 4 if family == "S" else 16
    eps_cpu = torch.randn(B, lat_c, side // 8, side // 8, generator=torch.Generator().manual_seed(99 + rank))
    pipe.vae.posterior_noise = eps_cpu.to(device)
    if family == "S":
        prompt = torch.randn(1, 77, 1024, generator=g).to(WDTYPE).to(device)

        def step():
            return pipe.sr_nhwc(lq, prompt, tile, overlap)
    else:
        from omgsr_amd.pipelines.omgsr_f import prepare_latent_image_ids
        prompt = torch.randn(1, 512, 4096, generator=g).to(WDTYPE).to(device)
        pooled = torch.randn(1, 768, generator=g).to(WDTYPE).to(device)
        text_ids = torch.zeros(512, 3, device=device, dtype=WDTYPE)
        image_ids = prepare_latent_image_ids(tile // 2, tile // 2, device, WDTYPE)

        def step():
            return pipe.sr_nhwc(lq, prompt, pooled, text_ids, image_ids, tile, overlap)

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
    elapsed = D.max_over_ranks(elapsed, device)
    images = B * world * args.steps
    value = images / elapsed

    roofline, extra = None, {}
    if rank == 0 and not args.no_roofline:
        lib = _lib.load()
        lib.omgsr_timing_reset(); lib.omgsr_timing_enable(1)
        with torch.no_grad():
            step()
        kinds = collect_roofline(_lib)
        lib.omgsr_timing_enable(0); lib.omgsr_timing_reset()
        ig = kinds.get(1)
        if ig and ig["ms"] > 0:
            ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "igemm_kernel (implicit-GEMM conv/linear, all launches of one step)",
                        "achieved": round(ach, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": None,
                        "launches": ig["launches"], "kernel_ms": round(ig["ms"], 3),
                        "algorithmic_tflop": round(ig["flops"] / 1e12, 3),
                        "algorithmic_bytes_per_launch": round(ig["bytes"] / ig["launches"])}
            # HBM bytes per launch of the same kernel family, from the committed rocprofv3 PMC passes of THIS workload
            # (FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction; tools/traffic_summary.py)
            tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
            if args.workload == "s1024" and tiled_vae and WDTYPE == torch.bfloat16 and B == 4 and os.path.isfile(tpath):
                fam = json.load(open(tpath))["families"].get("igemm")
                if fam:
                    roofline["traffic"] = round(fam["hbm_bytes_per_launch"])
                    roofline["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)"
        names = {1: "igemm", 2: "attention", 3: "groupnorm", 4: "layernorm", 5: "elementwise", 6: "softmax"}
        extra["kernel_ms_by_family"] = {names.get(k, str(k)): round(v["ms"], 3) for k, v in sorted(kinds.items())}
        extra["pipeline_frac_of_mfma_peak"] = round(tflop_per_img * B * args.steps / elapsed / PEAK_BF16_DENSE_TFLOPS, 4) if world == 1 else None

    cpu_baseline, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and family == "S":
        cpu_baseline, parity, oracle_img = cpu_leg(lq_cpu[:1], eps_cpu[:1], prompt.float().cpu(), out[:1], tile, overlap, side, tiled_vae)
        if WDTYPE == torch.bfloat16:
            # the same workload in the reference's other 16-bit --weight_dtype: same kernels (templates on the element type),
            # 8x finer mantissa; a short timed leg + parity of image 0 against the same oracle output
            extra["fp16_mode"] = fp16_leg(device, rank, world, tiled_vae, lq_cpu, eps_cpu, prompt, tile, overlap, oracle_img, B)

    if rank == 0:
        line = {
            "metric": "SR images/sec", "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if WDTYPE == torch.bfloat16 else "fp16", "data": "synthetic",
            "config": {"workload": f"OMGSR-{family} {side // 4}->{side}, batch={B}/GPU, {'bf16' if WDTYPE == torch.bfloat16 else 'fp16'}, seeded random weights at {'SD2.1-base' if family == 'S' else 'FLUX.1-dev'} shapes"
                                   + (", tiled VAE (VAEHook enc 256 / dec 64)" if tiled_vae else ""),
                       "global_batch": B * world, "latent_tile": tile, "tile_overlap": overlap, "mid_timestep": 273 if family == "S" else 244,
                       "parallelism": f"dp{world} (images sharded, RCCL weight broadcast {moved >> 20} MiB)"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
            "setup_s": round(build_s_secs, 1), **extra,
        }
        print(json.dumps(line))


def ops_nhwc(x_nchw):
    from omgsr_amd import ops
    return ops.nchw_to_nhwc(x_nchw.contiguous(), 8)


def cpu_leg(lq1, eps1, prompt, hip_out_nhwc, tile, overlap, side, tiled_vae=False):
    """fp32 CPU oracle on ONE image of the workload (bounded sample), and parity of the HIP output vs it."""
    from omgsr_amd import ops
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef, TiledVaeRef
    # 16 threads is the fastest eager-fp32 configuration on the GPU box's 2 x EPYC 9575F (measured with
    # tools/cpu_threads_probe.py: 16 thr 1.13 TFLOP/s, 32 thr 0.77, 64 thr 0.52, 128 thr 0.24)
    cores = int(os.environ.get("OMGSR_CPU_THREADS", min(16, os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    vae, unet = seeded_init_(R.AutoencoderKL(), 101).eval(), seeded_init_(R.UNet2DConditionModel(), 202).eval()
    vae.posterior_noise = eps1
    ref = OmgsrSRef(TiledVaeRef(vae, 256, 64) if tiled_vae else vae, unet, R.DDPMScheduler().alphas_cumprod[273], 273)
    with torch.no_grad():
        t0 = time.perf_counter()
        img = ref(lq1, prompt, tile, overlap)
        secs = time.perf_counter() - t0
    got = ops.nhwc_to_nchw(hip_out_nhwc.contiguous(), channels=3, dtype=torch.float32, clamp=(-1.0, 1.0)).cpu()
    parity = {"vs": "fp32 CPU oracle, same weights/inputs/eps, image 0", "rel_l2": round(rel_l2(got, img), 5),
              "psnr_db": round(psnr(got, img), 2)}
    base = {"value": round(1.0 / secs, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 image {side // 4}->{side}, fp32 eager PyTorch oracle, {secs:.1f} s, torch threads={torch.get_num_threads()}"}
    return base, parity, img


def fp16_leg(device, rank, world, tiled_vae, lq_cpu, eps_cpu, prompt, tile, overlap, oracle_img, B):
    global WDTYPE
    from omgsr_amd import ops
    from omgsr_amd.testing import psnr, rel_l2
    keep = WDTYPE
    try:
        WDTYPE = torch.float16
        ops.set_compute_dtype(torch.float16)
        pipe, _ = build_s(device, rank, world)
        if tiled_vae:
            pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
        pipe.vae.posterior_noise = eps_cpu.to(device)
        lq = ops_nhwc(lq_cpu.to(device))
        pr = prompt.to(torch.float16)
        with torch.no_grad():
            out = pipe.sr_nhwc(lq, pr, tile, overlap)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(2):
                out = pipe.sr_nhwc(lq, pr, tile, overlap)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 2
        got = ops.nhwc_to_nchw(out[:1].contiguous(), channels=3, dtype=torch.float32, clamp=(-1.0, 1.0)).cpu()
        return {"images_per_s": round(B / dt, 3), "ms_per_step": round(dt * 1e3, 3), "steps": 2,
                "rel_l2": round(rel_l2(got, oracle_img), 5), "psnr_db": round(psnr(got, oracle_img), 2)}
    finally:
        WDTYPE = keep
        ops.set_compute_dtype(keep)


if __name__ == "__main__":
    main()

"""Where a token GEMM's time goes by element kinds (GPU box): the UNet level-0 projection 147456 x 320 -> 320 and friends with
16-bit / split operands, 16-bit / fp32 outputs and residuals. Usage: python tools/bench_gemm_tiers.py [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
ops.set_compute_dtype(torch.float32)      # accurate tier: fp16 operands, fp32 streams allowed
SHAPES = [(147456, 320, 320), (147456, 320, 960), (147456, 1280, 320), (36864, 640, 640), (9216, 1280, 1280)]
FORMS = [(1, 1), (2, 1), (1, 2), (2, 2)]     # (operand terms, weight terms): 1, 2, 2 (wrapped), 3 K segments
if os.environ.get("FORMS"):
    FORMS = [tuple(int(c) for c in f) for f in os.environ["FORMS"].split(",")]
if os.environ.get("SHAPE"):
    SHAPES = [tuple(int(v) for v in os.environ["SHAPE"].split("x"))]
for M, K, N in SHAPES:
    w = (torch.randn(N, K, device=dev) / K ** 0.5)
    b = torch.zeros(N, device=dev)
    for split, w_split in FORMS:
        pw = ops.pack_linear_weight(w, b, split=split, w_split=w_split)
        x32 = torch.randn(1, M, K, device=dev) * 0.5
        x = ops.to_operand(x32, split)
        seg = split + (w_split == 2)          # K segments the GEMM contracts over
        for out_dtype, res_kind in ((ops.OUT_BF16, None), (ops.OUT_BF16, "f32"), (ops.OUT_F32, None), (ops.OUT_F32, "f32")):
            r = torch.randn(1, M, N, device=dev) if res_kind else None
            y = ops.linear(x, pw, residual=r, out_dtype=out_dtype)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                y = ops.linear(x, pw, residual=r, out_dtype=out_dtype)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / reps
            by = 2.0 * M * K * split + M * N * ((4 if out_dtype == ops.OUT_F32 else 2) + (4 if r is not None else 0)) + 2.0 * N * K * split
            print(f"M={M} K={K} N={N} split={split} w_split={w_split} ({seg} seg) out={'f32' if out_dtype == ops.OUT_F32 else 'f16'} res={res_kind}: {dt * 1e6:8.1f} us  "
                  f"{2.0 * M * K * N / dt / 1e12:7.1f} alg / {2.0 * M * K * N * seg / dt / 1e12:7.1f} executed TFLOP/s  {by / dt / 1e9:7.0f} GB/s", flush=True)

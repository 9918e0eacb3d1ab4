"""Micro-benchmark of the token-matrix GEMMs (UNet / DiT linears) at the default bench's sizes (GPU box).
Usage: OMGSR_IGEMM_MODE=reg|dma python tools/bench_linear.py [reps]
Prints TFLOP/s and the algorithmic HBM rate (x + out [+ residual] once, weights once)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops

SHAPES = [  # name, M, K, N, geglu, residual
    ("unet L0 q/k/v/out 320->320", 147456, 320, 320, False, True),
    ("unet L0 qk fused 320->640", 147456, 320, 640, False, False),
    ("unet L0 geglu 320->2x1280", 147456, 320, 1280, True, False),
    ("unet L0 ff out 1280->320", 147456, 1280, 320, False, True),
    ("unet L1 640->640", 36864, 640, 640, False, True),
    ("unet L1 geglu 640->2x2560", 36864, 640, 2560, True, False),
    ("unet L1 ff out 2560->640", 36864, 2560, 640, False, True),
    ("unet L2 1280->1280", 9216, 1280, 1280, False, True),
    ("unet L2 geglu 1280->2x5120", 9216, 1280, 5120, True, False),
    ("unet L2 ff out 5120->1280", 9216, 5120, 1280, False, True),
    ("flux 3072->3072 M=4608", 4608, 3072, 3072, False, True),
    ("flux 3072->9216 M=4608", 4608, 3072, 9216, False, False),
    ("small flux txt 3072->3072 M=512", 512, 3072, 3072, False, True),
    ("small flux txt 3072->12288 M=512", 512, 3072, 12288, False, False),
    ("small flux txt 12288->3072 M=512", 512, 12288, 3072, False, True),
    ("small unet512 1280->1280 M=2048", 2048, 1280, 1280, False, True),
    ("small unet512 640->640 M=8192", 8192, 640, 640, False, True),
    ("small unet512 320->320 M=32768", 32768, 320, 320, False, True),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
print("mode", os.environ.get("OMGSR_IGEMM_MODE", "auto"))
FILTER = os.environ.get("SHAPE_FILTER", "")
for name, M, K, N, geglu, res in [s for s in SHAPES if FILTER in s[0]]:
    x = (torch.randn(1, M, K, device=dev) * 0.5).to(ops.act_dtype())
    nw = 2 * N if geglu else N
    w = torch.randn(nw, K, device=dev) / K ** 0.5
    b = torch.zeros(nw, device=dev)
    pw = ops.pack_geglu_weight(w, b) if geglu else ops.pack_linear_weight(w, b)
    r = (torch.randn(1, M, N, device=dev) * 0.5).to(ops.act_dtype()) if res else None
    y = ops.linear(x, pw, residual=r)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        y = ops.linear(x, pw, residual=r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    fl = 2.0 * M * K * nw
    by = 2.0 * (M * K + M * N * (2 if res else 1) + nw * K)
    print(f"{name:32s} {dt * 1e6:9.1f} us  {fl / dt / 1e12:8.1f} TFLOP/s  {by / dt / 1e9:8.0f} GB/s   finite={bool(torch.isfinite(y.float()).all())}", flush=True)

"""us per launch of the accurate tier's mixed-precision 3x3 convs on the halo-tile kernel, one launch per shape: correction segments as fp8
(e4m3, per-tensor scales: split 3) against fp6 (e2m3, per-32-channel scales in the data: split 4, round 5), plus the two producers of the
operand (cast, GroupNorm apply + SiLU) in both forms. Usage: python tools/fp6_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops
dev = "cuda"
ops.set_compute_dtype(torch.float32)
g = torch.Generator().manual_seed(3)
SHAPES = [("VAE 128 -> 128, 4 x 1216 x 1216", 4, 1216, 1216, 128, 128), ("VAE 256 -> 256, 4 x 608 x 608", 4, 608, 608, 256, 256),
          ("VAE 512 -> 512, 4 x 304 x 304", 4, 304, 304, 512, 512), ("UNet 320 -> 320, 36 x 64 x 64", 36, 64, 64, 320, 320),
          ("UNet 640 -> 640, 36 x 32 x 32", 36, 32, 32, 640, 640)]
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6
for name, N, H, W, C, Co in SHAPES:
    xs = torch.randn(N, H, W, C, generator=g).mul_(0.5).to(dev)
    wt = torch.randn(Co, C, 3, 3, generator=g) * (9 * C) ** -0.5
    mean, rstd, _ = ops.group_norm_stats(xs, 32, 1e-6)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    row = f"{name:34s}"
    for split in (3, 4):
        x = ops.to_operand(xs, split)
        pc = ops.pack_conv_weight(wt, torch.zeros(Co), device=dev, split=split)
        us = timed(lambda: ops.conv2d(x, pc, pad=1))
        cast = timed(lambda: ops.to_operand(xs, split))
        gn = timed(lambda: ops.group_norm_apply(xs, mean, rstd, gam, bet, 32, ops.ACT_SILU, split=split))
        row += f" | {'fp8' if split == 3 else 'fp6'}: conv {us:8.1f} us ({2.0 * N * H * W * Co * 9 * C / us * 1e-6:6.1f} TF)  cast {cast:7.1f}  gn-apply {gn:7.1f}"
        del x
    print(row, flush=True)
    del xs

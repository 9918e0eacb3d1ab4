"""Where does a short-K token GEMM's time go? (GPU box) Sweeps N and K around the UNet's 147456 x 320 -> 320 problem in the bf16 tier and
fits time = a + b K per N: a = what a tile costs besides its K loop (prologue, epilogue, launch), b = the K loop.
Usage: python tools/bench_shortk.py [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
ops.set_compute_dtype(torch.bfloat16)


def run(M, K, N, res, out_f32=False):
    x = (torch.randn(1, M, K, device=dev) * 0.5).to(ops.act_dtype())
    pw = ops.pack_linear_weight(torch.randn(N, K, device=dev) / K ** 0.5, torch.zeros(N, device=dev))
    r = (torch.randn(1, M, N, device=dev) * 0.5).to(ops.act_dtype()) if res else None
    kw = dict(out_dtype=ops.OUT_F32) if out_f32 else {}
    for _ in range(3):
        ops.linear(x, pw, residual=r, **kw)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ops.linear(x, pw, residual=r, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


M = 147456
print("bf16 tier, M = 147456; us per launch (algorithmic GB/s incl. x, out, residual)")
for res in (False, True):
    for N in (128, 256, 320, 384, 640, 1280):
        row = []
        for K in (320, 640, 1280, 2560):
            us = run(M, K, N, res)
            by = 2.0 * (M * K + M * N * (2 if res else 1))
            row.append(f"K={K}: {us:7.1f} us ({by / us / 1e3:5.0f} GB/s, {2.0 * M * K * N / us / 1e6:6.0f} TF/s)")
        print(f"N={N:5d} residual={int(res)}  " + "  ".join(row), flush=True)
print("M sweep at K = 320, N = 320, residual:")
for Mx in (9216, 36864, 147456, 589824):
    us = run(Mx, 320, 320, True)
    print(f"M={Mx:7d}: {us:7.1f} us  {2.0 * (Mx * 320 * 3) / us / 1e3:5.0f} GB/s", flush=True)

"""Accurate-tier MX token GEMMs (igemm_gmx_kernel vs igemm_p8_kernel<MX>) under the ping-pong kernel's column-padding rule (GPU box):
    for P in 20 22 27; do OMGSR_P8_PAD_NUM=$P python tools/bench_mx_linear_pad.py; done"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
ops.set_compute_dtype(torch.float32)
for M, K, N in [(147456, 320, 320), (147456, 1280, 320), (147456, 320, 640), (36864, 640, 640), (36864, 2560, 640), (147456, 640, 320)]:
    w = torch.randn(N, K, device=dev) / K ** 0.5
    pw = ops.pack_linear_weight(w, torch.zeros(N, device=dev), split=3, w_split=2)
    x = ops.to_operand(torch.randn(1, M, K, device=dev) * 0.5, 3)
    r = torch.randn(1, M, N, device=dev)
    y = ops.linear(x, pw, residual=r, out_dtype=ops.OUT_F32)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        y = ops.linear(x, pw, residual=r, out_dtype=ops.OUT_F32)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print(f"pad {os.environ.get('OMGSR_P8_PAD_NUM', 'default')}: M={M} K={K} N={N} MX operand, fp32 out + residual: {dt * 1e6:8.1f} us  {2.0 * M * K * N / dt / 1e12:7.1f} TFLOP/s", flush=True)

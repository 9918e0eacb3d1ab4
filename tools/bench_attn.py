"""Micro-benchmark of the fused attention kernel at the pipelines' shapes (GPU box); also checks it against a torch fp32 softmax.
Usage: [OMGSR_ATTN_VARIANT=0] python tools/bench_attn.py [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops

SHAPES = [  # name, B, H, D, Lq, Lk
    ("flux joint 24 x 128, L 4608, B 1", 1, 24, 128, 4608, 4608),
    ("flux joint 24 x 128, L 4608, B 8", 8, 24, 128, 4608, 4608),
    ("unet self 5 x 64, L 4096, B 36", 36, 5, 64, 4096, 4096),
    ("unet self 10 x 64, L 1024, B 36", 36, 10, 64, 1024, 1024),
    ("unet self 20 x 64, L 256, B 36", 36, 20, 64, 256, 256),
    ("unet cross 5 x 64, Lk 77, B 36", 36, 5, 64, 4096, 77),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
dt_ = ops.act_dtype()
for name, B, H, D, Lq, Lk in SHAPES:
    q = (torch.randn(B, Lq, H * D, device=dev)).to(dt_)
    k = (torch.randn(B, Lk, H * D, device=dev)).to(dt_)
    ld = (Lk + 7) // 8 * 8
    v = torch.zeros(B, ld, H * D, device=dev)
    v[:, :Lk] = torch.randn(B, Lk, H * D, device=dev)
    vt = v.transpose(1, 2).contiguous().to(dt_)          # [B, H*D, ld]
    o = ops.attention(q, k, vt, H, D, D ** -0.5, Lk=Lk)
    # reference on one (batch, head)
    b, h = B - 1, H - 1
    qq = q[b, :, h * D:(h + 1) * D].float(); kk = k[b, :, h * D:(h + 1) * D].float(); vv = vt[b, h * D:(h + 1) * D, :Lk].float().t()
    ref = torch.softmax(qq @ kk.t() * D ** -0.5, dim=-1) @ vv
    err = ((o[b, :, h * D:(h + 1) * D].float() - ref).norm() / ref.norm()).item()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ops.attention(q, k, vt, H, D, D ** -0.5, Lk=Lk, out=o)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    fl = 4.0 * B * H * Lq * Lk * D
    print(f"{name:36s} {dt * 1e3:8.3f} ms  {fl / dt / 1e12:8.1f} TFLOP/s   rel err {err:.2e}", flush=True)

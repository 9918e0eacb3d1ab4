"""GroupNorm apply in the accurate tier (fp32 in -> fp16 / split out) against plain streaming references (GPU box):
how far is the kernel from what a copy of the same bytes achieves?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops
ops.set_compute_dtype(torch.float32)
dev = "cuda"
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
for name, N, H, W, C in [("vae 128ch 1024^2 b4", 4, 1024, 1024, 128), ("vae 256ch 512^2 b4", 4, 512, 512, 256), ("vae 512ch 256^2 b4", 4, 256, 256, 512),
                         ("unet 320ch 64^2 b36", 36, 64, 64, 320)]:
    x = torch.randn(N, H, W, C, device=dev)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    m, r, _ = ops.group_norm_stats(x, 32, 1e-6)
    y16 = torch.empty_like(x, dtype=torch.float16)
    rows = []
    for label, fn, bytes_ in [
        ("apply + SiLU  f32 -> f16", lambda: ops.group_norm_apply(x, m, r, g, b, 32, ops.ACT_SILU), 6),
        ("apply (no act) f32 -> f16", lambda: ops.group_norm_apply(x, m, r, g, b, 32, ops.ACT_NONE), 6),
        ("apply + SiLU  f32 -> split", lambda: ops.group_norm_apply(x, m, r, g, b, 32, ops.ACT_SILU, split=2), 8),
        ("torch copy_   f32 -> f16", lambda: y16.copy_(x), 6),
        ("torch clone   f32 -> f32", lambda: x.clone(), 8),
    ]:
        dt = timeit(fn)
        rows.append(f"{label:28s} {dt*1e3:7.3f} ms {x.numel()*bytes_/dt/1e12:5.2f} TB/s")
    print(name); [print("   ", r_) for r_ in rows]

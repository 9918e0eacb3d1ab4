"""GroupNorm kernel bandwidth on OMGSR shapes (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops
dev = "cuda"
for name, N, H, W, C in [("vae 128ch 512^2 b8", 8, 512, 512, 128), ("vae 256ch 256^2 b8", 8, 256, 256, 256), ("vae 512ch 128^2 b8", 8, 128, 128, 512),
                         ("vae 128ch 1024^2 b4", 4, 1024, 1024, 128), ("tile 128ch 688x688 b4", 4, 688, 688, 128), ("unet 320ch 64^2 b36", 36, 64, 64, 320)]:
    x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    for fn, label, mult in [(lambda: ops.group_norm_stats(x, 32, 1e-6), "stats", 1), (None, "apply", 2)]:
        if fn is None:
            m, r, _ = ops.group_norm_stats(x, 32, 1e-6)
            fn = lambda: ops.group_norm_apply(x, m, r, g, b, 32, ops.ACT_SILU)
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        print(f"{name:26s} {label:6s} {dt*1e3:7.3f} ms  {x.numel()*2*mult/dt/1e12:6.2f} TB/s", flush=True)

print("LayerNorm")
for name, rows, C in [("unet L0 [147456, 320]", 147456, 320), ("unet L1 [36864, 640]", 36864, 640), ("unet L2 [9216, 1280]", 9216, 1280),
                      ("flux [4608, 3072]", 4608, 3072)]:
    x = torch.randn(1, rows, C, device=dev).to(torch.bfloat16)
    a = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    ops.layer_norm(x, a, b, 1e-5); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): ops.layer_norm(x, a, b, 1e-5)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print(f"{name:26s} {dt*1e6:8.1f} us  {x.numel()*4/dt/1e12:6.2f} TB/s", flush=True)

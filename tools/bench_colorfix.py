"""SURVEY §8(f) f1 measurement: the uint8 / colour-fix kernels at the default bench's image size vs the HBM roofline,
with the oracle (the reference's CPU algorithm) timed beside it. Usage (GPU box): python tools/bench_colorfix.py [B] [side]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops
from omgsr_amd.colorfix import color_fix, image_to_model_input
from oracle import colorfix_ref as R

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator().manual_seed(0)
sr = (torch.rand(B, S, S, 8, generator=g) * 2.2 - 1.1).to(ops.act_dtype()).cuda()
src = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8).cuda()
px = B * S * S
# algorithmic HBM bytes per pixel: 16-B NHWC pixel read per pass, 3-B source read, 3-B result write; wavelet: 2 x 3 fp32 planes
# written once + per level (read 2x3, write 2x3, high read+write 3) + final
ALG = {"nofix": 16 + 3, "adain": 16 + 3 + 16 + 3, "wavelet": 16 + 3 + 24 + 5 * (24 + 24 + 24) + 24 + 3}
for m in ("nofix", "adain", "wavelet"):
    out = color_fix(sr, src, m); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        out = color_fix(sr, src, m)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print(f"{m:8s} {dt * 1e6:9.1f} us / {B} images   {B / dt:9.0f} images/s   {px * ALG[m] / dt / 1e9:7.0f} GB/s algorithmic ({ALG[m]} B/px) = {px * ALG[m] / dt / 8e12 * 100:4.1f} % of 8 TB/s")
x = image_to_model_input(src); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    x = image_to_model_input(src)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 20
print(f"to-input {dt * 1e6:9.1f} us / {B} images   {px * 19 / dt / 1e9:7.0f} GB/s")
# CPU: the oracle = the reference's algorithm (torch fp32 on the host), one image
torch.set_num_threads(int(os.environ.get("OMGSR_CPU_THREADS", "16")))
sr1 = sr[:1, ..., :3].permute(0, 3, 1, 2).cpu()
s1 = src[:1].permute(0, 3, 1, 2).cpu()
for m, f in (("adain", R.adain_color_fix_u8), ("wavelet", R.wavelet_color_fix_u8)):
    t = time.perf_counter()
    for _ in range(3):
        r = f(R.model_output_to_u8(sr1), s1)
    dt = (time.perf_counter() - t) / 3
    print(f"cpu {m:8s} {dt * 1e3:8.1f} ms / image ({torch.get_num_threads()} threads)   {1 / dt:7.1f} images/s")

"""Micro-benchmark of the implicit-GEMM kernels on the OMGSR layer shapes (GPU box).
Usage: [TIER=fp32] [RES=1] [SHAPE_FILTER=...] OMGSR_IGEMM_MODE=reg|dma python tools/bench_igemm.py [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops

SHAPES = [  # name, N, H, W, Cin, Cout, ksize
    ("vae 128->128 @512", 8, 512, 512, 128, 128, 3),
    ("vae 256->256 @256", 8, 256, 256, 256, 256, 3),
    ("vae 512->512 @128", 8, 128, 128, 512, 512, 3),
    ("vae 512->512 @64", 8, 64, 64, 512, 512, 3),
    ("unet 320->320 @64", 8, 64, 64, 320, 320, 3),
    ("unet 640->640 @32", 8, 32, 32, 640, 640, 3),
    ("unet 1280->1280 @16", 8, 16, 16, 1280, 1280, 3),
    ("unet lin 320->320 M=32768", 1, 1, 32768, 320, 320, 1),
    ("unet lin 640->5120 M=8192", 1, 1, 8192, 640, 5120, 1),
    ("flux lin 3072->3072 M=4608", 1, 1, 4608, 3072, 3072, 1),
    ("flux lin 3072->12288 M=4608", 1, 1, 4608, 3072, 12288, 1),
    ("flux lin 15360->3072 M=4608", 1, 1, 4608, 15360, 3072, 1),
    ("flux ctx lin 3072->3072 M=4096", 1, 1, 4096, 3072, 3072, 1),
    ("flux ctx lin 12288->3072 M=4096", 1, 1, 4096, 12288, 3072, 1),
    ("flux ctx lin 3072->6144 M=4096", 1, 1, 4096, 3072, 6144, 1),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda"
TIER = os.environ.get("TIER", "")              # "fp32" = accurate tier (fp32 stream tensors, fp16 operands)
RES = os.environ.get("RES", "") == "1"         # add a residual in the stream type (a ResnetBlock's conv2 / an attention to_out)
if TIER:
    ops.set_compute_dtype({"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[TIER])
print("mode", os.environ.get("OMGSR_IGEMM_MODE", "auto"))
FILTER = os.environ.get("SHAPE_FILTER", "")
for name, N, H, W, Cin, Cout, k in [s for s in SHAPES if FILTER in s[0]]:
    x = (torch.randn(N, H, W, Cin, device=dev) * 0.5).to(ops.act_dtype())
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    pw = ops.pack_conv_weight(w, torch.zeros(Cout, device=dev))
    pad = 1 if k == 3 else 0
    r = (torch.randn(N, H, W, Cout, device=dev) * 0.5).to(ops.stream_dtype()) if RES else None
    y = ops.conv2d(x, pw, pad=pad, residual=r)
    if RES:      # check against the unfused form
        y0 = ops.conv2d(x, pw, pad=pad).float() + r.float()
        err = ((y.float() - y0).norm() / y0.norm()).item()
        assert err < 5e-3, err
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        y = ops.conv2d(x, pw, pad=pad, residual=r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    fl = 2.0 * N * H * W * Cin * k * k * Cout
    chk = ""
    if k == 3 and os.environ.get("CHECK_CONV"):      # against torch's fp32 conv on the rounded operands (one image)
        wq = w.to(ops.act_dtype()).float()
        ref = torch.nn.functional.conv2d(x[:1].float().permute(0, 3, 1, 2), wq, padding=1).permute(0, 2, 3, 1)
        if RES:
            ref = ref + r[:1].float()
        chk = f"  rel err vs torch {((y[:1].float() - ref).norm() / ref.norm()).item():.2e}"
    if k == 1:       # GEMM-shaped: check against a torch fp32 matmul on the rounded operands
        wq = w.reshape(Cout, Cin).to(ops.act_dtype()).float()
        ref = x.reshape(-1, Cin).float() @ wq.t() + (r.reshape(-1, Cout).float() if RES else 0.0)
        chk = f"  rel err vs torch {((y.reshape(-1, Cout).float() - ref).norm() / ref.norm()).item():.2e}"
    print(f"{name:32s} {dt * 1e3:8.3f} ms  {fl / dt / 1e12:8.1f} TFLOP/s   finite={bool(torch.isfinite(y.float()).all())}{chk}", flush=True)

"""Host-side probe: oracle VAE decode time vs torch thread count (picks bench.py's cpu_baseline threads)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import diffusers_ref as R
from omgsr_amd.testing import seeded_init_
vae = seeded_init_(R.AutoencoderKL(), 101).eval()
z = torch.randn(1, 4, 32, 32)
print("cpu_count", os.cpu_count())
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    torch.set_num_threads(n)
    with torch.no_grad():
        vae.decode(z)
        t = time.perf_counter(); vae.decode(z); dt = time.perf_counter() - t
    print(f"threads={n}: decode 32x32 latent {dt:.2f}s  ({0.63 / dt:.2f} TFLOP/s)", flush=True)

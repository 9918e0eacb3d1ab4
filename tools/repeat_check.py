"""Bit-repeatability of the whole default-bench step (OMGSR-S 256->1024, tiled VAE, batch 4): N identical steps must
produce identical bits (any LDS / global race shows up as a handful of differing pixels). Usage: python tools/repeat_check.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from omgsr_amd import ops
from omgsr_amd.testing import synthetic_lq
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
pipe, _ = bench.build_s(dev, 0, 1)
pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
lq = ops.nchw_to_nhwc(synthetic_lq(4, 1024, 1024, seed=1).to(dev), 8)
pipe.vae.posterior_noise = torch.randn(4, 4, 128, 128, generator=torch.Generator().manual_seed(2)).to(dev)
prompt = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).to(dev)
with torch.no_grad():
    ref = pipe.sr_nhwc(lq, prompt, 64, 32)
    bad = 0
    for i in range(n):
        cur = pipe.sr_nhwc(lq, prompt, 64, 32)
        if not torch.equal(cur, ref):
            d = (cur.float() - ref.float()).abs()
            print(f"repeat {i}: {int((d > 0).sum())} elements differ, max {d.max().item():.4f}")
            bad += 1
print(f"{bad} of {n} repeats differ")
sys.exit(1 if bad else 0)

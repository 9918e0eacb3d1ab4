import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from omgsr_amd.testing import synthetic_lq
from omgsr_amd import ops
dev = torch.device("cuda", 0)
pipe, _ = bench.build_s(dev, 0, 1)
pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
lq = ops.nchw_to_nhwc(synthetic_lq(4, 1024, 1024, seed=1).to(dev), 8)
pipe.vae.posterior_noise = torch.randn(4, 4, 128, 128).to(dev)
prompt = torch.randn(1, 77, 1024).to(torch.bfloat16).to(dev)
with torch.no_grad():
    for _ in range(2): pipe.sr_nhwc(lq, prompt, 64, 32)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter(); pipe.sr_nhwc(lq, prompt, 64, 32); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")

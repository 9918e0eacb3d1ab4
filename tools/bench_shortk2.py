"""Round 5: the UNet's short-K token GEMMs (147456 rows) under the LDS-DMA kernel's row-tile heights (OMGSR_DMA_BM=64|128|256 in the environment)
and numbers of 128-column tiles. us per launch, bf16 tier. Usage: OMGSR_DMA_BM=128 python tools/bench_shortk2.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops
dev = "cuda"
ops.set_compute_dtype(torch.bfloat16)
M = 147456
def run(K, N, res, reps=30):
    x = (torch.randn(1, M, K, device=dev) * 0.5).to(ops.act_dtype())
    pw = ops.pack_linear_weight(torch.randn(N, K, device=dev) / K ** 0.5, torch.zeros(N, device=dev))
    r = (torch.randn(1, M, N, device=dev) * 0.5).to(ops.act_dtype()) if res else None
    for _ in range(3):
        ops.linear(x, pw, residual=r)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ops.linear(x, pw, residual=r)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6
print("BM", os.environ.get("OMGSR_DMA_BM", "auto"))
for res in (0, 1):
    for K in (320, 640):
        print(f"res={res} K={K}: " + "  ".join(f"N={N}: {run(K, N, res):6.1f}" for N in (128, 256, 320, 384, 640)), flush=True)

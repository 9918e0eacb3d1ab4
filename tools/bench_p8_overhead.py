"""Fixed cost per 256 x 256 tile of igemm_p8_kernel (GPU box): the Flux token GEMMs at M = 36864 for a sweep of K with the epilogue's options,
time per launch -> us per tile round = a + b * (K / 64). `a` is what one workgroup per CU cannot hide: dispatch + prologue + epilogue.
    OMGSR_P8_MIN_K=128 python tools/bench_p8_overhead.py [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMGSR_P8_MIN_K", "128")
from omgsr_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
M = 36864
for N, res, gate in ((3072, True, True), (3072, False, False), (12288, False, False)):
    rows = []
    for K in (128, 256, 512, 1024, 3072, 6144):
        x = (torch.randn(1, M, K, device=dev) * 0.5).to(ops.act_dtype())
        w = torch.randn(N, K, device=dev) / K ** 0.5
        pw = ops.pack_linear_weight(w, torch.zeros(N, device=dev))
        r = (torch.randn(1, M, N, device=dev) * 0.5).to(ops.act_dtype()) if res else None
        g = torch.rand(N, device=dev) if gate else None
        y = ops.linear(x, pw, residual=r, gate=g, out_dtype=ops.OUT_BF16)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            y = ops.linear(x, pw, residual=r, gate=g, out_dtype=ops.OUT_BF16)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        tiles = (M // 256) * (N // 256)
        per_round = dt * 1e6 / (tiles / 256)
        rows.append((K, dt * 1e6, per_round, 2.0 * M * K * N / dt / 1e12))
    (k0, _, p0, _), (k1, _, p1, _) = rows[-2], rows[-1]
    b = (p1 - p0) / ((k1 - k0) / 64)
    print(f"N={N} residual={res} gate={gate}: us per K-tile {b:.3f} ({256 * 256 * 64 * 2 * 256 / b / 1e6:.0f} TFLOP/s in the loop)")
    for K, us, pr, tf in rows:
        print(f"   K={K:5d}  {us:9.1f} us/launch  {pr:7.2f} us per round of 256 tiles  fixed {pr - b * K / 64:6.2f} us  {tf:7.1f} TFLOP/s")

"""Soak: the same launch repeated many times must give the same bits (a mis-placed LDS-DMA wait shows up as rare wrong tiles).
Kernels with hand-placed waits: igemm_p8_kernel (one-tile and persistent forms), attn_kernel (LDS-DMA path), igemm_halo_kernel, igemm_dma_kernel, igemm_gmx_kernel. Usage: soak_repeat.py [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = "cuda"
g = torch.Generator().manual_seed(7)
def soak(name, fn, n):
    # (operand tensors in the MX forms are raw bytes behind a 16-bit dtype - some are NaN patterns: compare bit patterns, not values)
    bits = lambda t: t.view(torch.int16) if t.dtype in (torch.float16, torch.bfloat16) else t       # noqa: E731
    ref = bits(fn()); bad = 0
    for _ in range(n):
        bad += int(not torch.equal(bits(fn()), ref))
    print(f"{name:48s} {n} repeats, {bad} differ", flush=True)
    return bad
total = 0
for tier in (torch.bfloat16, torch.float32):
    ops.set_compute_dtype(tier)
    dt = ops.act_dtype()
    tag = "accurate" if tier == torch.float32 else "bf16"
    for M, K, N in [(4608, 3072, 3072), (36864, 3072, 12288), (4096, 12288, 3072), (25700, 1536, 512)]:
        x = (torch.randn(1, M, K, generator=g) * 0.5).to(dt).to(dev)
        pw = ops.pack_linear_weight(torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g), device=dev)
        total += soak(f"[{tag}] p8 linear {M}x{K}->{N}", lambda: ops.linear(x, pw, out_dtype=ops.OUT_BF16), reps if M < 30000 else reps // 4)
    for B, H, D, L in [(2, 24, 128, 4608), (8, 5, 64, 4096)]:
        qk = (torch.randn(B, L, 2 * H * D, generator=g) * 0.5).to(dt).to(dev)
        vt = (torch.randn(B, H * D, L, generator=g) * 0.5).to(dt).to(dev)
        total += soak(f"[{tag}] attention {B}x{H}x{D} L {L}", lambda: ops.attention(qk, qk, vt, H, D, D ** -0.5, q_col=0, k_col=H * D, Lk=L), reps)
    for N_, C, Co, H_, W_ in [(4, 128, 128, 256, 256), (4, 512, 512, 64, 64), (36, 320, 320, 64, 64)]:
        xc = (torch.randn(N_, H_, W_, C, generator=g) * 0.5).to(dt).to(dev)
        pc = ops.pack_conv_weight(torch.randn(Co, C, 3, 3, generator=g) * (9 * C) ** -0.5, torch.zeros(Co), device=dev)
        total += soak(f"[{tag}] halo conv {N_}x{H_}x{W_} {C}->{Co}", lambda: ops.conv2d(xc, pc, pad=1), reps)
    x2 = (torch.randn(1, 147456, 320, generator=g) * 0.5).to(dt).to(dev)
    p2 = ops.pack_linear_weight(torch.randn(320, 320, generator=g) * 320 ** -0.5, None, device=dev)
    total += soak(f"[{tag}] dma linear 147456x320->320", lambda: ops.linear(x2, p2, out_dtype=ops.OUT_BF16), reps)
# the mixed-precision GEMM (accurate tier only: fp16 K-steps then block-scaled fp8 K-steps behind a run-time ring stage)
ops.set_compute_dtype(torch.float32)
for M, C, N in [(147456, 320, 320), (36864, 640, 2560), (9216, 1280, 1280)]:
    xm = ops.to_operand((torch.randn(1, M, C, generator=g) * 0.5).to(dev), 3)
    pm = ops.pack_linear_weight(torch.randn(N, C, generator=g) * C ** -0.5, torch.randn(N, generator=g), device=dev, split=3)
    total += soak(f"[accurate] MX linear {M}x{C}->{N} (gmx | p8-MX)", lambda: ops.linear(xm, pm), reps)
# round 5: the small-M forms of the same kernels (64 / 128-row tiles) and the split-K launch group of the halo-tile kernel (one image per call)
for M, C, N in [(1024, 640, 640), (256, 1280, 1280), (4096, 320, 320)]:
    xm = ops.to_operand((torch.randn(1, M, C, generator=g) * 0.5).to(dev), 3)
    pm = ops.pack_linear_weight(torch.randn(N, C, generator=g) * C ** -0.5, torch.randn(N, generator=g), device=dev, split=3)
    total += soak(f"[accurate] MX linear {M}x{C}->{N} (gmx, small tiles)", lambda: ops.linear(xm, pm), reps)
for N_, C, Co, H_, W_ in [(1, 512, 512, 64, 64), (1, 640, 640, 32, 32), (1, 320, 320, 64, 64)]:
    xc = ops.to_operand((torch.randn(N_, H_, W_, C, generator=g) * 0.5).to(dev), 3)
    pc = ops.pack_conv_weight(torch.randn(Co, C, 3, 3, generator=g) * (9 * C) ** -0.5, torch.zeros(Co), device=dev, split=3)
    rr = torch.randn(N_, H_, W_, Co, generator=g).to(dev)
    total += soak(f"[accurate] MX halo conv, split-K {N_}x{H_}x{W_} {C}->{Co}", lambda: ops.conv2d(xc, pc, pad=1, residual=rr), reps)
# ... the fp6 correction chunks (OMGSR_EL_MX6): spatial, FLAT (22- and 27-piece patch) and split-K forms, and the quad-cooperative producers
for N_, C, Co, H_, W_ in [(4, 128, 128, 256, 256), (4, 256, 512, 75, 75), (8, 512, 512, 40, 40), (4, 256, 256, 38, 38), (1, 512, 512, 64, 64), (36, 320, 320, 64, 64)]:
    xf = (torch.randn(N_, H_, W_, C, generator=g) * 0.5).to(dev)
    pc = ops.pack_conv_weight(torch.randn(Co, C, 3, 3, generator=g) * (9 * C) ** -0.5, torch.zeros(Co), device=dev, split=4)
    xc = ops.to_operand(xf, 4)
    total += soak(f"[accurate] fp6 halo conv {N_}x{H_}x{W_} {C}->{Co}", lambda: ops.conv2d(xc, pc, pad=1), reps)
    total += soak(f"[accurate] fp6 cast {N_}x{H_}x{W_}x{C}", lambda: ops.to_operand(xf, 4), reps // 4)
    mean, rstd, _ = ops.group_norm_stats(xf, 32, 1e-6)
    total += soak(f"[accurate] fp6 GroupNorm apply {N_}x{H_}x{W_}x{C}", lambda: ops.group_norm_apply(xf, mean, rstd, None, None, 32, ops.ACT_SILU, split=4), reps // 4)
# ... the up-sampler pair in the fp6 form: a conv whose epilogue writes the fp6 operand (OUT6 instantiations, plain and fp6 operand) and the phase-form conv that reads it
for split in (1, 4):
    xo = ops.to_operand((torch.randn(4, 96, 128, 128, generator=g) * 0.5).to(dev), split)
    po = ops.pack_conv_weight(torch.randn(128, 128, 3, 3, generator=g) * (9 * 128) ** -0.5, torch.zeros(128), device=dev, split=split)
    ro = torch.randn(4, 96, 128, 128, generator=g).to(dev)
    total += soak(f"[accurate] OUT6 halo conv (operand split {split}) 4x96x128 128->128", lambda: ops.conv2d(xo, po, pad=1, residual=ro, out_dtype=ops.OUT_BF16, out_split=4), reps)
xu = ops.to_operand((torch.randn(4, 86, 86, 256, generator=g) * 0.5).to(dev), 4)
pu = ops.pack_conv_weight(torch.randn(256, 256, 3, 3, generator=g) * (9 * 256) ** -0.5, torch.zeros(256), device=dev, split=4, upsample_phases=True)
total += soak("[accurate] fp6 phase-form up-sampling conv 4x86x86 256->256", lambda: ops.conv2d(xu, pu, pad=1, upsample=True), reps)
# ... and GroupNorm apply + SiLU as the conv's patch producer (fast tiers: a wave normalises its own LDS-DMA'd pieces in place)
for tier in (torch.bfloat16, torch.float16):
    ops.set_compute_dtype(tier)
    dt = ops.act_dtype()
    for N_, C, Co, H_, W_ in [(4, 128, 128, 256, 256), (12, 512, 128, 64, 64), (2, 128, 8, 256, 256)]:
        xs = (torch.randn(N_, H_, W_, C, generator=g) * 1.5).to(dt).to(dev)
        pc = ops.pack_conv_weight(torch.randn(Co, C, 3, 3, generator=g) * (9 * C) ** -0.5, torch.zeros(Co), device=dev, cout_multiple=8)
        spec = ops.GnSpec(torch.randn(N_, 32, generator=g).to(dev) * 0.1, (1.0 + 0.1 * torch.randn(N_, 32, generator=g)).abs().to(dev),
                          (1.0 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev), 32, ops.ACT_SILU)
        total += soak(f"[{dt}] GN-fused halo conv {N_}x{H_}x{W_} {C}->{Co}", lambda: ops.conv2d(xs, pc, pad=1, gn=spec), reps)
# round 6: the flash kernel with q / k / P / V as two-term splits (K_lo and V^T_lo tiles ride through LDS behind the hi tiles: more LDS-DMA pieces per
# tile under the same vmcnt(0) wait) - the range-fallback tier's UNet attention, LDS-DMA and register-staged paths; and the XCD-ordered 1-D grid
ops.set_compute_dtype(torch.float32, operand_dtype=torch.bfloat16)
for B, H, D, Lq, Lk in [(8, 5, 64, 4096, 4096), (36, 5, 64, 4096, 77), (4, 10, 64, 1024, 1024)]:
    inner = H * D
    sp = lambda t: torch.cat([t.to(torch.bfloat16), (t - t.to(torch.bfloat16).float()).to(torch.bfloat16)], -1)       # noqa: E731
    qq = sp(torch.randn(B, Lq, inner, generator=g)).to(dev)
    kk = sp(torch.randn(1 if Lk == 77 else B, Lk, inner, generator=g)).to(dev)
    vts = ops.transpose_split((torch.randn(1 if Lk == 77 else B, Lk, inner, generator=g)).to(dev), (Lk + 7) // 8 * 8)
    total += soak(f"[fallback] split attention {B}x{H}x{D} Lq {Lq} Lk {Lk}", lambda: ops.attention(qq, kk, vts, H, D, D ** -0.5, Lk=Lk, out_split=2, q_lo_col=inner, k_lo_col=inner), reps)
ops.set_compute_dtype(torch.bfloat16)
print("TOTAL differing:", total)
sys.exit(1 if total else 0)

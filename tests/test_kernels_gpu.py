"""Kernel-level parity: every C-ABI entry point vs a plain PyTorch fp32 reference of the same op
(computed on the CPU from the same bf16-rounded inputs). Tolerances: the kernels accumulate in fp32
and round once to bf16, so the bound is bf16 output rounding (2^-9 relative) plus fp32
accumulation-order noise."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from omgsr_amd import ops
    return ops


@pytest.fixture(autouse=True, params=["bf16", "fp16"])
def compute_dtype(request):
    """Every kernel test runs in both 16-bit modes (omgsr_set_compute_dtype); inputs are bf16-representable
    values, which fp16 holds exactly in the normal range, so one fp32 reference serves both."""
    ops = _ops()
    ops.set_compute_dtype(torch.bfloat16 if request.param == "bf16" else torch.float16)
    yield request.param
    ops.set_compute_dtype(torch.bfloat16)


def bf(x):
    return x.to(_ops().act_dtype())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()


def assert_close(got, ref, name, rel_l2=4e-3, max_ulps=3.0):
    got = got.float().cpu()
    ref = ref.float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).norm() / ref.norm().clamp_min(1e-12)
    # per-element bound: a few bf16 ulps of max(|ref|, typical magnitude)
    scale = ref.abs().clamp_min(ref.abs().mean())
    worst = ((got - ref).abs() / scale).max().item()
    assert err.item() < rel_l2, f"{name}: rel-L2 {err.item():.3e} (worst elt {worst:.3e})"
    assert worst < max_ulps * 2 ** -8, f"{name}: worst element error {worst:.3e} (rel-L2 {err.item():.3e})"


def nhwc(x):  # NCHW f32 -> NHWC bf16 on device
    return bf(x.permute(0, 2, 3, 1).contiguous()).to(DEV)


def to_nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2)


CONV_CASES = [
    # N, Cin, Cout, H, W, stride, pad(t,b,l,r), upsample, bias, residual, act
    (1, 32, 32, 8, 8, 1, (1, 1, 1, 1), False, True, False, 0),
    (2, 128, 128, 32, 32, 1, (1, 1, 1, 1), False, True, True, 0),
    (1, 320, 640, 16, 16, 1, (1, 1, 1, 1), False, True, False, 1),
    (2, 8, 128, 24, 40, 1, (1, 1, 1, 1), False, True, False, 0),       # conv_in (3 -> padded 8)
    (1, 128, 3, 32, 32, 1, (1, 1, 1, 1), False, True, False, 0),        # conv_out
    (1, 64, 8, 16, 16, 1, (1, 1, 1, 1), False, True, False, 0),
    (2, 128, 128, 32, 32, 2, (0, 1, 0, 1), False, True, False, 0),      # VAE downsample: pad(0,1,0,1), s2
    (1, 320, 320, 16, 16, 2, (1, 1, 1, 1), False, True, False, 0),      # UNet downsample: p1, s2
    (1, 256, 256, 12, 20, 1, (1, 1, 1, 1), True, True, False, 0),       # nearest-2x folded into the gather
    (1, 40, 72, 9, 7, 1, (1, 1, 1, 1), False, False, False, 0),         # ragged everything
    (3, 640, 1280, 8, 8, 1, (1, 1, 1, 1), False, True, True, 0),
    # ragged spatial sizes (tiled-VAE tiles): exercise the halo-tile kernel's border handling
    (1, 64, 128, 86, 43, 1, (1, 1, 1, 1), False, True, False, 0),
    (2, 128, 256, 20, 50, 1, (1, 1, 1, 1), False, True, True, 1),
    (1, 32, 128, 9, 33, 1, (1, 1, 1, 1), False, False, False, 0),
    (2, 256, 128, 64, 64, 1, (1, 1, 1, 1), False, True, True, 0),
    (2, 128, 128, 64, 96, 1, (1, 1, 1, 1), True, True, False, 0),       # 2x upsample on the halo-tile path (>= 192 tiles)
    (1, 64, 128, 43, 150, 1, (1, 1, 1, 1), True, False, False, 0),     # ... ragged
    (2, 128, 3, 128, 192, 1, (1, 1, 1, 1), False, True, False, 0),     # conv_out on the narrow halo shape (Cout <= 32, >= 192 tiles)
    (4, 128, 8, 86, 150, 1, (1, 1, 1, 1), False, True, False, 0),
    (1, 64, 16, 256, 200, 1, (1, 1, 1, 1), False, False, True, 1),
    (2, 32, 32, 100, 260, 1, (1, 1, 1, 1), True, True, False, 0),
    # convs that miss the halo kernel and take the ping-pong GEMM kernel's implicit-GEMM form (K >= 1536, >= 128 tiles of 256 x 256)
    (72, 256, 512, 16, 16, 1, (1, 1, 1, 1), False, True, True, 0),     # 16-wide maps (UNet 16 x 16 level)
    (8, 256, 256, 128, 128, 2, (0, 1, 0, 1), False, True, False, 0),   # VAE down-sampling conv
    (9, 192, 256, 118, 122, 2, (1, 1, 1, 1), False, True, False, 1),   # stride 2, ragged, 33 792 rows (last tile partial)
    (144, 256, 256, 8, 8, 1, (1, 1, 1, 1), True, False, False, 0),     # nearest-2x upsampling to 16 x 16
    (40, 512, 512, 40, 40, 1, (1, 1, 1, 1), False, True, True, 0),     # 40-wide maps (tiled-VAE encoder's last level)
    # narrow maps on the halo kernel's FLAT form (256 consecutive positions of the flattened padded map per workgroup)
    (30, 128, 256, 38, 38, 1, (1, 1, 1, 1), False, True, True, 1),
    (64, 128, 128, 36, 45, 1, (1, 1, 1, 1), False, False, False, 0),
    (100, 64, 128, 20, 20, 1, (1, 1, 1, 1), False, True, True, 0),     # a 32-position run spans three image rows
    (40, 96, 160, 37, 41, 1, (1, 1, 1, 1), False, True, False, 0),     # ragged Cout (two column tiles, the second 32 wide)
    # ... with the 27-piece patch (maps 46-80 pixels wide)
    (16, 128, 256, 75, 75, 1, (1, 1, 1, 1), False, True, True, 1),
    (8, 128, 128, 64, 80, 1, (1, 1, 1, 1), False, False, False, 0),
    (12, 64, 128, 50, 46, 1, (1, 1, 1, 1), False, True, True, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3(case):
    ops = _ops()
    N, Cin, Cout, H, W, stride, pad, ups, use_bias, use_res, act = case
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    b = rnd(Cout, seed=3) if use_bias else None
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    xin = F.pad(xin, (pad[2], pad[3], pad[0], pad[1]))
    ref = F.conv2d(xin, w, b, stride=stride)
    if act == 1:
        ref = F.silu(ref)
    res = None
    if use_res:
        res = rnd(*ref.shape, seed=4)
        ref = ref + res
    pw = ops.pack_conv_weight(w, b, device=DEV)
    y = ops.conv2d(nhwc(x), pw, stride=stride, pad=pad, upsample=ups, act=act,
                   residual=None if res is None else nhwc(res))
    assert_close(to_nchw(y), ref, f"conv{case}")


# N, Cin, Cout, H, W (low-res), bias, act: nearest-2x upsampling convs on the phase-decomposed form (four 2 x 2 convolutions of the
# low-res map with phase-summed kernels, igemm_halo_kernel<.., 4>): decoder / UNet upsamplers, ragged tiled-VAE tile sizes
UPS_PHASE_CASES = [(2, 128, 128, 64, 96, True, 0), (1, 64, 128, 43, 150, False, 0), (4, 512, 512, 32, 32, True, 0), (1, 256, 256, 86, 86, True, 1),
                   (2, 512, 512, 43, 43, True, 0), (9, 640, 640, 32, 32, True, 0), (1, 256, 96, 19, 40, False, 0)]


@pytest.mark.parametrize("case", UPS_PHASE_CASES)
def test_upsample_conv_phase_form(case, monkeypatch):
    ops = _ops()
    N, Cin, Cout, H, W, use_bias, act = case
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    b = rnd(Cout, seed=3) if use_bias else None
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    if act == 1:
        ref = F.silu(ref)
    pw = ops.pack_conv_weight(w, b, device=DEV, upsample_phases=True)
    assert pw.w_ph is not None and tuple(pw.w_ph.shape) == (4, Cin // 32, 4, pw.cout_pad, 32)
    y = ops.conv2d(nhwc(x), pw, pad=1, upsample=True, act=act, gn_groups=32)
    assert tuple(y.shape) == (N, 2 * H, 2 * W, Cout)
    # (the weights of taps that share an input pixel are summed BEFORE the one rounding to the compute type: the worst element sits a
    # little further out than with nine separately rounded taps, the rel-L2 is the same)
    assert_close(to_nchw(y), ref, f"upsample-phase{case}", max_ulps=4.0)
    # the GroupNorm statistics describe the stored tensor (left by the epilogue, one slot per wave tile and phase, when the problem is
    # large enough for the phase form; the small cases fall back to the gather form and the stand-alone statistics pass)
    big = N * ((W + 31) // 32) * ((H + 7) // 8) * ((Cout + 127) // 128) * 4 >= 192
    assert (getattr(y, "_omgsr_gn", None) is not None) or not big
    mean, rstd, var = ops.group_norm_stats(y, 32, 1e-6)
    r = y.float().cpu().reshape(N, 4 * H * W, 32, Cout // 32)
    assert torch.allclose(mean.cpu(), r.mean(dim=(1, 3)), atol=2e-3, rtol=2e-3)
    assert torch.allclose(var.cpu(), r.var(dim=(1, 3), unbiased=False), atol=2e-3, rtol=5e-3)
    # ... and the gather form (nine taps on the virtual map) agrees to the output rounding
    pw9 = ops.pack_conv_weight(w, b, device=DEV)
    y9 = ops.conv2d(nhwc(x), pw9, pad=1, upsample=True, act=act)
    assert_close(to_nchw(y), to_nchw(y9), f"phase vs gather {case}", max_ulps=6.0)     # two independently rounded 16-bit results


# shape groups of one tiled-VAE layer (rows, height, width) sharing one weight: one launch over all of them (omgsr_igemm_multi)
MULTI_CASES = [
    # Cin, Cout, upsample, residual, groups
    (128, 128, False, True, [(36, 40, 40), (12, 40, 32), (12, 32, 40), (4, 32, 32)]),        # encoder groups at a deep level
    (512, 512, False, False, [(4, 86, 86), (4, 86, 64), (4, 64, 86), (4, 64, 64)]),          # decoder groups, first level
    (256, 256, True, False, [(4, 43, 43), (4, 43, 32), (4, 32, 43), (4, 32, 32)]),           # upsampling conv: phase form, merged
    (128, 3, False, False, [(2, 96, 80), (2, 64, 80), (1, 64, 64)]),                         # conv_out: narrow shape
    (64, 128, False, False, [(1, 9, 33), (2, 16, 16)]),                                      # too small for the halo kernel even together
]


@pytest.mark.parametrize("case", MULTI_CASES)
def test_conv_multi_launch(case):
    ops = _ops()
    Cin, Cout, ups, use_res, groups = case
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    b = rnd(Cout, seed=3)
    pw = ops.pack_conv_weight(w, b, device=DEV, cout_multiple=8, upsample_phases=ups)
    xs, refs, ress = [], [], []
    for i, (n, h, wd) in enumerate(groups):
        x = rnd(n, Cin, h, wd, seed=10 + i)
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
        ref = F.conv2d(xin, w, b, padding=1)
        res = rnd(*ref.shape, seed=20 + i) if use_res else None
        xs.append(nhwc(x)); ress.append(None if res is None else nhwc(res)); refs.append(ref if res is None else ref + res)
    gn = 32 if Cout % 32 == 0 else 0
    ys = ops.conv2d_multi(xs, pw, pad=1, upsample=ups, residuals=ress if use_res else None, gn_groups=gn)
    singles = [ops.conv2d(x, pw, pad=1, upsample=ups, residual=r, gn_groups=gn) for x, r in zip(xs, ress)]
    for y, y1, ref, (n, h, wd) in zip(ys, singles, refs, groups):
        assert_close(to_nchw(y)[:, :Cout], ref, f"multi{case[:4]}{(n, h, wd)}", max_ulps=4.0 if ups else 3.0)
        assert_close(to_nchw(y), to_nchw(y1), f"multi vs single {(n, h, wd)}", max_ulps=6.0)
        if gn:      # statistics (fused by whichever kernel ran, or the stand-alone pass) describe the stored tensor
            mean, rstd, var = ops.group_norm_stats(y, gn, 1e-6)
            r = y.float().cpu().reshape(y.shape[0], -1, gn, Cout // gn)
            assert torch.allclose(mean.cpu(), r.mean(dim=(1, 3)), atol=2e-3, rtol=2e-3)
            assert torch.allclose(var.cpu(), r.var(dim=(1, 3), unbiased=False), atol=2e-3, rtol=5e-3)


@pytest.mark.parametrize("M,K,Nout", [(64, 320, 320), (4096, 320, 960), (1000, 1280, 1280), (77, 1024, 640), (300, 64, 3072),
                                      # round 5, small-M regime (one image per call): the LDS-DMA kernel's 64- / 128-row tiles
                                      (128, 1280, 1280), (512, 640, 640), (65, 256, 10240), (2048, 320, 320), (33, 1280, 320)])
def test_linear(M, K, Nout):
    ops = _ops()
    x = rnd(2, M, K, seed=5)
    w = rnd(Nout, K, seed=6, scale=1.0 / math.sqrt(K))
    b = rnd(Nout, seed=7)
    ref = F.linear(x, w, b)
    pw = ops.pack_linear_weight(w, b, device=DEV)
    y = ops.linear(bf(x).to(DEV), pw)
    assert_close(y, ref, f"linear{(M, K, Nout)}")


def test_linear_epilogues():
    ops = _ops()
    M, K, Nout = 333, 256, 384
    x = rnd(1, M, K, seed=8)
    w = rnd(Nout, K, seed=9, scale=1.0 / math.sqrt(K))
    b = rnd(Nout, seed=10)
    gate = rnd(Nout, seed=11)
    res = rnd(1, M, Nout, seed=12)
    pw = ops.pack_linear_weight(w, b, device=DEV)
    y = ops.linear(bf(x).to(DEV), pw, act=ops.ACT_GELU_TANH, gate=gate.to(DEV), residual=bf(res).to(DEV))
    ref = res + gate * F.gelu(F.linear(x, w, b), approximate="tanh")
    assert_close(y, ref, "gelu_tanh+gate+residual")
    y32 = ops.linear(bf(x).to(DEV), pw, out_dtype=ops.OUT_F32, alpha=0.125)
    ref32 = 0.125 * F.linear(x, w) + b
    assert y32.dtype == torch.float32
    assert_close(y32, ref32, "f32 out + alpha", rel_l2=1e-5, max_ulps=0.05)


def test_geglu():
    ops = _ops()
    M, K, inner = 200, 320, 1280
    x = rnd(1, M, K, seed=13)
    w = rnd(2 * inner, K, seed=14, scale=1.0 / math.sqrt(K))
    b = rnd(2 * inner, seed=15, scale=0.1)
    h = F.linear(x, w, b)
    a, g = h.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    pw = ops.pack_geglu_weight(w, b, device=DEV)
    y = ops.linear(bf(x).to(DEV), pw)
    assert y.shape[-1] == inner
    assert_close(y, ref, "geglu")


def test_linear_transposed_out():
    ops = _ops()
    B, L, K, Nout = 2, 77, 1024, 320
    x = rnd(B, L, K, seed=16)
    w = rnd(Nout, K, seed=17, scale=1.0 / math.sqrt(K))
    pw = ops.pack_linear_weight(w, None, device=DEV)
    yt = ops.linear_t(bf(x).to(DEV), pw, L)
    assert yt.shape == (B, Nout, 80)
    ref = F.linear(x, w).transpose(1, 2)
    assert_close(yt[:, :, :L], ref, "linear_t")
    assert (yt[:, :, L:] == 0).all()


@pytest.mark.parametrize("B,L,K,Nout", [(2, 256, 320, 320), (3, 4096, 320, 320), (1, 4608, 3072, 3072), (2, 96, 640, 200)])
def test_linear_transposed_out_whole_blocks(B, L, K, Nout):
    """L % 32 == 0: the column-wise 16-byte epilogue path of the transposed output (register-staged, LDS-DMA and ping-pong kernels),
    with a bias; and through linear_t_into at a key offset of a wider V^T buffer."""
    ops = _ops()
    x = rnd(B, L, K, seed=116)
    w = rnd(Nout, K, seed=117, scale=1.0 / math.sqrt(K))
    b = rnd(Nout, seed=118)
    pw = ops.pack_linear_weight(w, b, device=DEV)
    xd = bf(x).to(DEV)
    yt = ops.linear_t(xd, pw, L)
    assert yt.shape == (B, Nout, L)
    ref = F.linear(x, w, b).transpose(1, 2)
    assert_close(yt, ref, f"linear_t whole blocks {(B, L, K, Nout)}")
    vt = torch.full((B, Nout, L + 64), 7.0, device=DEV, dtype=ops.act_dtype())
    ops.linear_t_into(xd, pw, vt, 32)
    assert_close(vt[:, :, 32:32 + L], ref, "linear_t_into at key 32")
    assert (vt[:, :, :32] == 7).all() and (vt[:, :, 32 + L:] == 7).all()


def test_split_k_paths():
    """Small-M / long-K problems are split over K (fp32 partial tiles + reduce pass with the full epilogue)."""
    ops = _ops()
    M, K, Nout = 300, 3072, 384
    x = rnd(1, M, K, seed=70)
    w = rnd(Nout, K, seed=71, scale=K ** -0.5)
    b, gate, res = rnd(Nout, seed=72), rnd(Nout, seed=73), rnd(1, M, Nout, seed=74)
    pw = ops.pack_linear_weight(w, b, device=DEV)
    from omgsr_amd import _lib
    import ctypes
    y = ops.linear(bf(x).to(DEV), pw, act=ops.ACT_GELU_TANH, gate=gate.to(DEV), residual=bf(res).to(DEV))
    assert_close(y, res + gate * F.gelu(F.linear(x, w, b), approximate="tanh"), "split-K gelu+gate+residual")
    y32 = ops.linear(bf(x).to(DEV), pw, out_dtype=ops.OUT_F32, alpha=0.5)
    assert_close(y32, 0.5 * F.linear(x, w) + b, "split-K f32", rel_l2=1e-5, max_ulps=0.05)
    inner = 640
    wg = rnd(2 * inner, 2048, seed=75, scale=2048 ** -0.5)
    bg = rnd(2 * inner, seed=76, scale=0.1)
    xg = rnd(1, 200, 2048, seed=77)
    h = F.linear(xg, wg, bg)
    a_, g_ = h.chunk(2, dim=-1)
    assert_close(ops.linear(bf(xg).to(DEV), ops.pack_geglu_weight(wg, bg, device=DEV)), a_ * F.gelu(g_), "split-K geglu")
    # the policy really splits these shapes
    a = _lib.IgemmArgs()
    a.N, a.Ho, a.Wo, a.Cin, a.Cout, a.K_pad, a.batch, a.act, a.out_layout = 1, 1, M, K, Nout, K, 1, 0, 0
    assert _lib.load().omgsr_igemm_workspace_bytes(ctypes.byref(a)) > 0


def test_linear_into_slices():
    """Projections written straight into slices of joint buffers (Flux joint sequence / [attn|mlp] concat)."""
    ops = _ops()
    Lc, Li, K, Nout = 24, 200, 256, 384
    xc, xi = rnd(Lc, K, seed=60), rnd(Li, K, seed=61)
    wc, wi = rnd(Nout, K, seed=62, scale=K ** -0.5), rnd(Nout, K, seed=63, scale=K ** -0.5)
    bc, bi = rnd(Nout, seed=64), rnd(Nout, seed=65)
    joint = torch.full((Lc + Li, 2 * Nout + 64), 7.0, dtype=_ops().act_dtype(), device=DEV)
    ops.linear_into(bf(xc).to(DEV), ops.pack_linear_weight(wc, bc, device=DEV), joint, 0, Nout)
    ops.linear_into(bf(xi).to(DEV), ops.pack_linear_weight(wi, bi, device=DEV), joint, Lc, Nout, act=ops.ACT_GELU_TANH)
    j = joint.float().cpu()
    assert_close(j[:Lc, Nout:2 * Nout], F.linear(xc, wc, bc), "linear_into ctx")
    assert_close(j[Lc:, Nout:2 * Nout], F.gelu(F.linear(xi, wi, bi), approximate="tanh"), "linear_into img")
    assert (j[:, :Nout] == 7).all() and (j[:, 2 * Nout:] == 7).all()          # nothing outside the slice is touched
    vt = torch.zeros((Nout, Lc + Li), dtype=_ops().act_dtype(), device=DEV)
    ops.linear_t_into(bf(xc).to(DEV), ops.pack_linear_weight(wc, bc, device=DEV), vt, 0)
    ops.linear_t_into(bf(xi).to(DEV), ops.pack_linear_weight(wi, bi, device=DEV), vt, Lc)
    ref = torch.cat([F.linear(xc, wc, bc), F.linear(xi, wi, bi)], 0).t()
    assert_close(vt, ref, "linear_t_into")


def test_bmm_nt():
    ops = _ops()
    B, M, K, Nn = 2, 200, 512, 256
    a = rnd(B, M, K, seed=18)
    b = rnd(B, Nn, K, seed=19)
    s = ops.bmm_nt(bf(a).to(DEV), bf(b).to(DEV), alpha=K ** -0.5, out_dtype=ops.OUT_F32)
    ref = torch.einsum("bmk,bnk->bmn", a, b) * K ** -0.5
    assert_close(s, ref, "bmm_nt", rel_l2=1e-5, max_ulps=0.05)


@pytest.mark.parametrize("N,C,H,W,act", [(2, 128, 64, 64, 1), (1, 320, 16, 16, 0), (2, 1920, 8, 8, 1), (1, 512, 24, 40, 1), (1, 2560, 8, 8, 1)])
def test_group_norm(N, C, H, W, act):
    ops = _ops()
    x = bf(rnd(N, C, H, W, seed=20) * 2.0 + 0.5).float()
    gamma = rnd(C, seed=21) + 1.0
    beta = rnd(C, seed=22)
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    if act:
        ref = F.silu(ref)
    xd = nhwc(x)
    mean, rstd, var = ops.group_norm_stats(xd, 32, 1e-6)
    xr = x.reshape(N, 32, -1)
    assert_close(mean, xr.mean(-1), "gn mean", rel_l2=1e-5, max_ulps=0.01)
    assert_close(var, xr.var(-1, unbiased=False), "gn var", rel_l2=1e-4, max_ulps=0.05)
    y = ops.group_norm_apply(xd, mean, rstd, gamma.to(DEV), beta.to(DEV), 32, act)
    assert_close(to_nchw(y), ref, f"group_norm{(N, C, H, W)}")


@pytest.mark.parametrize("rows,C", [(100, 320), (64, 640), (50, 1280), (33, 3072)])
def test_layer_norm(rows, C):
    ops = _ops()
    x = bf(rnd(rows, C, seed=23) * 3.0 + 1.0).float()
    a = rnd(C, seed=24) + 1.0
    b = rnd(C, seed=25)
    ref = F.layer_norm(x, (C,), a, b, eps=1e-5)
    y = ops.layer_norm(bf(x).to(DEV), a.to(DEV), b.to(DEV), 1e-5)
    assert_close(y, ref, f"layer_norm{(rows, C)}")
    y0 = ops.layer_norm(bf(x).to(DEV), None, None, 1e-6)
    assert_close(y0, F.layer_norm(x, (C,), eps=1e-6), "layer_norm no affine")


def _sdpa_ref(q, k, v, heads, scale):
    B, Lq, inner = q.shape
    D = inner // heads
    qh = q.reshape(B, Lq, heads, D).transpose(1, 2)
    kh = k.reshape(k.shape[0], -1, heads, D).transpose(1, 2)
    vh = v.reshape(v.shape[0], -1, heads, D).transpose(1, 2)
    s = torch.einsum("bhqd,bhkd->bhqk", qh, kh.expand(B, -1, -1, -1)) * scale
    o = torch.einsum("bhqk,bhkd->bhqd", s.softmax(-1), vh.expand(B, -1, -1, -1))
    return o.transpose(1, 2).reshape(B, Lq, inner)


@pytest.mark.parametrize("B,H,D,Lq,Lk,bcast", [
    (2, 5, 64, 256, 256, False), (1, 10, 64, 1024, 1024, False), (2, 20, 64, 64, 64, False),
    (2, 5, 64, 256, 77, True), (1, 5, 64, 100, 77, False), (1, 4, 128, 320, 320, False), (1, 2, 128, 200, 136, False),
    (1, 2, 128, 200, 192, False), (2, 5, 64, 256, 128, True), (1, 3, 128, 1000, 4608, False)])        # LDS-DMA staging (Lk % 64 == 0): ragged Lq, broadcast K / V, a long sweep
def test_attention(B, H, D, Lq, Lk, bcast):
    ops = _ops()
    inner = H * D
    q = rnd(B, Lq, inner, seed=26)
    Bk = 1 if bcast else B
    k = rnd(Bk, Lk, inner, seed=27)
    v = rnd(Bk, Lk, inner, seed=28)
    scale = D ** -0.5
    ref = _sdpa_ref(q, k, v, H, scale)
    ld = (Lk + 7) // 8 * 8
    vt = torch.zeros(Bk, inner, ld)
    vt[:, :, :Lk] = v.transpose(1, 2)
    o = ops.attention(bf(q).to(DEV), bf(k).to(DEV), bf(vt).to(DEV), H, D, scale, Lk=Lk)
    # P is rounded to bf16 before the PV product -> a little looser than a GEMM
    assert_close(o, ref, f"attention{(B, H, D, Lq, Lk)}", rel_l2=6e-3, max_ulps=6.0)


def test_attention_spiked_max():
    """Force the online-softmax rescale: one key dominates late in the sweep."""
    ops = _ops()
    B, H, D, L = 1, 1, 64, 512
    q = rnd(B, L, D, seed=29)
    k = rnd(B, L, D, seed=30)
    v = rnd(B, L, D, seed=31)
    k[0, 300] = q[0, 5] * 4.0      # spikes q row 5 in the 5th key tile
    k[0, 500] = q[0, 70] * 6.0
    k[0, 400] = q[0, 9] * 0.6      # a late maximum BELOW the deferred-rescale threshold (2^8 in the scaled base-2 domain): p > 1, no rescale
    k = bf(k).float()
    scale = D ** -0.5
    ref = _sdpa_ref(q, k, v, H, scale)
    o = ops.attention(bf(q).to(DEV), bf(k).to(DEV), bf(v.transpose(1, 2).contiguous()).to(DEV), H, D, scale)
    assert_close(o, ref, "attention spiked", rel_l2=6e-3, max_ulps=6.0)


def test_attention_strided_fused_qkv():
    """q/k read straight out of a fused [q|k] projection buffer with column offsets."""
    ops = _ops()
    B, H, D, L = 2, 5, 64, 192
    inner = H * D
    qk = rnd(B, L, 2 * inner, seed=32)
    v = rnd(B, L, inner, seed=33)
    scale = D ** -0.5
    ref = _sdpa_ref(qk[..., :inner], qk[..., inner:], v, H, scale)
    qk_d = bf(qk).to(DEV)
    o = ops.attention(qk_d, qk_d, bf(v.transpose(1, 2).contiguous()).to(DEV), H, D, scale, q_col=0, k_col=inner)
    assert_close(o, ref, "attention fused qk", rel_l2=6e-3, max_ulps=6.0)


def test_softmax_rows():
    ops = _ops()
    s = torch.randn(37, 4096, generator=torch.Generator().manual_seed(34)) * 4
    p = ops.softmax_rows(s.to(DEV))
    assert_close(p, s.softmax(-1), "softmax_rows")
    s2 = torch.randn(3, 16384, generator=torch.Generator().manual_seed(35)) * 4
    assert_close(ops.softmax_rows(s2.to(DEV)), s2.softmax(-1), "softmax_rows 16k")
    s3 = torch.randn(5, 1664, generator=torch.Generator().manual_seed(36)) * 4
    p3 = ops.softmax_rows(s3.to(DEV), valid=1600)
    assert_close(p3[:, :1600], s3[:, :1600].softmax(-1), "softmax_rows masked")
    assert (p3[:, 1600:] == 0).all()


def test_rmsnorm_rope():
    ops = _ops()
    B, L, H, D = 2, 96, 3, 128
    x = rnd(B, L, 2 * H * D, seed=36)
    w = rnd(H, D, seed=37) + 1.0          # per-head weight rows (q heads / k heads of a fused buffer)
    pos = torch.arange(L + 8, dtype=torch.float64)
    freqs = 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float64) / D))
    ang = torch.outer(pos, freqs)
    cos = ang.cos().repeat_interleave(2, dim=1).float()
    sin = ang.sin().repeat_interleave(2, dim=1).float()
    col0, pos0 = H * D, 8
    xs = x[..., col0:].reshape(B, L, H, D)
    xn = xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-6) * w[None, None]
    c = cos[pos0:pos0 + L][None, :, None, :]
    s = sin[pos0:pos0 + L][None, :, None, :]
    xr = torch.stack([-xn[..., 1::2], xn[..., 0::2]], dim=-1).flatten(-2)
    ref = x.clone()
    ref[..., col0:] = (xn * c + xr * s).reshape(B, L, H * D)
    xd = bf(x).to(DEV)
    ops.rmsnorm_rope_(xd, w.to(DEV), cos.to(DEV), sin.to(DEV), H, D, col0=col0, pos0=pos0)
    assert_close(xd, ref, "rmsnorm_rope")


def test_layout_and_latent_ops():
    ops = _ops()
    x = rnd(2, 3, 20, 12, seed=38)
    xh = ops.nchw_to_nhwc(x.to(DEV))
    assert xh.shape == (2, 20, 12, 8)
    assert torch.equal(xh[..., :3].float().cpu(), x.permute(0, 2, 3, 1))
    assert (xh[..., 3:] == 0).all()
    back = ops.nhwc_to_nchw(xh, channels=3, dtype=torch.float32, clamp=(-1.0, 1.0))
    assert torch.equal(back.cpu(), x.clamp(-1, 1))
    a, b = rnd(2, 5, 7, 16, seed=39), rnd(2, 5, 7, 24, seed=40)
    cat = ops.concat_channels(bf(a).to(DEV), bf(b).to(DEV))
    assert torch.equal(cat.float().cpu(), torch.cat([a, b], -1))
    # posterior sample
    mom = rnd(2, 6, 5, 8, seed=41)
    eps = torch.randn(2, 6, 5, 4, generator=torch.Generator().manual_seed(42))
    z = ops.vae_sample(bf(mom).to(DEV), eps.to(DEV), 4, 0.1159, 0.3611)
    mu, lv = mom[..., :4], mom[..., 4:].clamp(-30, 20)
    ref = ((mu + torch.exp(0.5 * lv) * eps) - 0.1159) * 0.3611
    assert_close(z[..., :4], ref, "vae_sample")
    assert (z[..., 4:] == 0).all()
    # axpby
    u, v = rnd(1000, seed=43), rnd(1000, seed=44)
    r = ops.axpby(bf(u).to(DEV), bf(v).to(DEV), 1.0 / 0.797, -0.6035 / 0.797, 0.0, 1.0 / 0.18215)
    assert_close(r, (u / 0.797 - v * 0.6035 / 0.797) / 0.18215, "axpby")
    # crop
    t = rnd(2, 9, 11, 8, seed=45)
    cr = ops.crop_nhwc(bf(t).to(DEV), 2, 3, 4, 5)
    assert torch.equal(cr.float().cpu(), t[:, 2:6, 3:8])
    # flux pack / unpack
    lat = rnd(2, 8, 12, 16, seed=46)  # NHWC
    tok = ops.flux_pack(bf(lat).to(DEV), 16)
    nchw = lat.permute(0, 3, 1, 2)
    ref_tok = nchw.reshape(2, 16, 4, 2, 6, 2).permute(0, 2, 4, 1, 3, 5).reshape(2, 24, 64)
    assert torch.equal(tok.float().cpu(), ref_tok)
    un = ops.flux_unpack(tok, 8, 12)
    assert torch.equal(un.float().cpu(), lat)


def test_tile_stitch_ops():
    ops = _ops()
    N, H, W, Cc, th, tw = 2, 12, 10, 4, 8, 8
    acc = torch.zeros(N, H, W, Cc, device=DEV)
    wsum = torch.zeros(1, H, W, 1, device=DEV)
    wt = torch.rand(th, tw, generator=torch.Generator().manual_seed(47)) + 0.1
    ref_acc = torch.zeros(N, H, W, Cc)
    ref_w = torch.zeros(H, W)
    for i, (y0, x0) in enumerate([(0, 0), (4, 2), (4, 0)]):
        tile = rnd(N, th, tw, 8, seed=48 + i)
        ops.tile_accumulate(bf(tile).to(DEV), wt.to(DEV), acc, y0, x0)
        ops.tile_accumulate(None, wt.to(DEV), wsum, y0, x0)
        ref_acc[:, y0:y0 + th, x0:x0 + tw] += tile[..., :Cc] * wt[None, :, :, None]
        ref_w[y0:y0 + th, x0:x0 + tw] += wt
    assert torch.allclose(acc.cpu(), ref_acc, atol=1e-6)
    ref_w = ref_w.clamp_min(1e-3)
    wsum.clamp_(min=1e-3)
    out = ops.tile_normalise(acc, wsum)
    assert_close(out[..., :Cc], ref_acc / ref_w[None, :, :, None], "tile_normalise")


@pytest.mark.parametrize("N,C,Cout,H,W,res", [(2, 128, 128, 128, 192, False), (2, 256, 256, 80, 160, True), (2, 512, 512, 64, 96, True),
                                               (2, 128, 256, 86, 150, False), (40, 128, 256, 38, 40, True), (16, 128, 256, 75, 72, True)])      # >= 192 halo tiles each; the last on the FLAT form
def test_conv_fused_groupnorm_statistics(N, C, Cout, H, W, res):
    """omgsr_igemm's gn_partial: the conv epilogue emits the (sum, sum of squares) of what it stores; GroupNorm of
    the result must equal F.group_norm of the conv output, and must not launch the statistics read pass."""
    ops = _ops()
    x = rnd(N, C, H, W, seed=90)
    w = rnd(Cout, C, 3, 3, seed=91, scale=(9 * C) ** -0.5)
    b = rnd(Cout, seed=92)
    r = rnd(N, Cout, H, W, seed=93) if res else None
    ref_conv = F.conv2d(x, w, b, padding=1) + (r if res else 0)
    gamma, beta = rnd(Cout, seed=94) + 1.0, rnd(Cout, seed=95)
    ref = F.silu(F.group_norm(ref_conv, 32, gamma, beta, eps=1e-6))
    pw = ops.pack_conv_weight(w, b, device=DEV)
    y = ops.conv2d(nhwc(x), pw, pad=1, residual=None if r is None else nhwc(r), gn_groups=32)
    assert getattr(y, "_omgsr_gn", None) is not None, "the halo path should have emitted statistics for this shape"
    assert_close(to_nchw(y), ref_conv, "conv out")
    mean, rstd, var = ops.group_norm_stats(y, 32, 1e-6)
    g = ref_conv.view(N, 32, -1)
    assert torch.allclose(mean.cpu(), g.mean(-1), atol=2e-3, rtol=2e-3)
    assert torch.allclose(var.cpu(), g.var(-1, unbiased=False), atol=2e-3, rtol=4e-3)
    out = ops.group_norm_apply(y, mean, rstd, gamma.to(DEV), beta.to(DEV), 32, ops.ACT_SILU)
    assert_close(to_nchw(out), ref, "fused-stat groupnorm", rel_l2=6e-3, max_ulps=6.0)
    # same tensor without the handle: the classic two-kernel path agrees to fp32 rounding of the statistics
    y2 = y.clone()
    m2, r2, v2 = ops.group_norm_stats(y2, 32, 1e-6)
    assert torch.allclose(mean, m2, atol=2e-3, rtol=2e-3) and torch.allclose(var, v2, atol=2e-3, rtol=4e-3)


def test_groupnorm_merged_tile_statistics():
    """omgsr_groupnorm_finalize_merged + apply_shared: the tiled VAE's pixel-weighted merge of per-tile (mean, var)
    (infer/vaehook.py GroupNormParam.summary) in one launch, rows (tile, image) tile-major."""
    ops = _ops()
    N, C, G = 2, 64, 32
    shapes = {(20, 24): 3, (20, 9): 2, (7, 9): 1}           # (h, w) -> tiles
    tens, tiles = [], []
    for i, ((h, w), tcount) in enumerate(shapes.items()):
        tens.append((rnd(tcount * N, C, h, w, seed=100 + i) * (1.0 + i) + 0.25 * i).to(torch.bfloat16).float())
        tiles.append(tcount)
    tot = sum(t.shape[2] * t.shape[3] * tiles[k] for k, t in enumerate(tens))
    mean_ref = torch.zeros(N, G); var_ref = torch.zeros(N, G)
    for k, t in enumerate(tens):
        g = t.view(tiles[k], N, G, -1)
        wgt = t.shape[2] * t.shape[3] / tot
        mean_ref += g.mean(-1).sum(0) * wgt
        var_ref += g.var(-1, unbiased=False).sum(0) * wgt
    dev = [nhwc(t) for t in tens]
    mean, rstd, var = ops.group_norm_stats_merged(dev, tiles, N, G, 1e-6)
    assert torch.allclose(mean.cpu(), mean_ref, atol=1e-4, rtol=1e-4)
    assert torch.allclose(var.cpu(), var_ref, atol=1e-4, rtol=1e-4)
    assert torch.allclose(rstd.cpu(), torch.rsqrt(var_ref + 1e-6), rtol=1e-4)
    gamma, beta = rnd(C, seed=110) + 1.0, rnd(C, seed=111)
    for k, t in enumerate(tens):
        y = ops.group_norm_apply_shared(dev[k], mean, rstd, gamma.to(DEV), beta.to(DEV), G, ops.ACT_SILU)
        m = mean_ref.repeat(tiles[k], 1)[:, :, None, None].repeat_interleave(C // G, 1)
        r = torch.rsqrt(var_ref + 1e-6).repeat(tiles[k], 1)[:, :, None, None].repeat_interleave(C // G, 1)
        ref = F.silu((t - m) * r * gamma[None, :, None, None] + beta[None, :, None, None])
        assert_close(to_nchw(y), ref, f"shared-stat apply group {k}")


def _threshold_cases():
    """Shapes that straddle the dispatcher's decision points (192 / 256 / 512 tiles, split-K length, 32-pixel halo width,
    192-row tiles), with and without residual / fused statistics: every path must agree with torch."""
    import random
    rng = random.Random(2024)
    cases = []
    for _ in range(28):
        C = rng.choice([32, 64, 128, 256, 320, 512, 640])
        Cout = rng.choice([96, 128, 256, 320, 384, 512])
        target_tiles = rng.choice([150, 190, 200, 250, 260, 500, 520, 700])
        ntn = (Cout + 127) // 128
        px = max(256, target_tiles * 256 // ntn)
        W = rng.choice([16, 24, 32, 40, 48, 64, 86, 96])
        N = rng.choice([1, 2, 3])
        H = max(8, px // (W * N))
        if 2.0 * N * H * W * C * 9 * Cout > 6e10:           # keep the CPU reference to a second or two
            continue
        cases.append((N, C, Cout, H, W, rng.random() < 0.5, rng.random() < 0.5))
    return cases


@pytest.mark.parametrize("case", _threshold_cases())
def test_conv_shapes_around_dispatch_thresholds(case):
    ops = _ops()
    N, C, Cout, H, W, use_res, want_stats = case
    x = rnd(N, C, H, W, seed=200)
    w = rnd(Cout, C, 3, 3, seed=201, scale=(9 * C) ** -0.5)
    b = rnd(Cout, seed=202)
    r = rnd(N, Cout, H, W, seed=203) if use_res else None
    ref = F.conv2d(x, w, b, padding=1) + (r if use_res else 0)
    pw = ops.pack_conv_weight(w, b, device=DEV)
    y = ops.conv2d(nhwc(x), pw, pad=1, residual=None if r is None else nhwc(r), gn_groups=32 if want_stats and Cout % 32 == 0 else 0)
    assert_close(to_nchw(y), ref, f"conv{case}")
    if want_stats and Cout % 32 == 0:
        mean, rstd, var = ops.group_norm_stats(y, 32, 1e-6)          # fused partials when the path left them, else the read pass
        g = ref.view(N, 32, -1)
        assert torch.allclose(mean.cpu(), g.mean(-1), atol=3e-3, rtol=3e-3), f"stats{case}"
        assert torch.allclose(var.cpu(), g.var(-1, unbiased=False), atol=3e-3, rtol=6e-3), f"stats{case}"


@pytest.mark.parametrize("M,K,Nout", [(9216, 1280, 1280), (9216, 640, 320), (4608, 3072, 768), (2304, 2560, 1280), (36864, 320, 640),
                                       (12288, 512, 384), (50000, 320, 320)])
def test_linear_shapes_around_dispatch_thresholds(M, K, Nout):
    ops = _ops()
    x = rnd(1, M, K, seed=210)
    w = rnd(Nout, K, seed=211, scale=K ** -0.5)
    b = rnd(Nout, seed=212)
    res = rnd(1, M, Nout, seed=213)
    pw = ops.pack_linear_weight(w, b, device=DEV)
    y = ops.linear(bf(x).to(DEV), pw, residual=bf(res).to(DEV))
    assert_close(y, F.linear(x, w, b) + res, f"linear{(M, K, Nout)}")


def test_ping_pong_gemm_kernel():
    """igemm_p8_kernel (256 x 256 x 64, two wave groups half a phase apart; GEMM-shaped problems with K >= 1536 and >= 200 tiles):
    a ragged M (rows past M are clamped and dropped), bias + residual, GEGLU, f32 output, a batched bmm, and the same launch
    repeated (a mis-placed LDS-DMA wait shows up as rare wrong tiles, not as a wrong mean)."""
    ops = _ops()
    M, K, Nout = 25700, 1536, 512                      # 101 x 2 tiles of 256 x 256
    x = rnd(1, M, K, seed=301)
    w = rnd(Nout, K, seed=302, scale=K ** -0.5)
    b = rnd(Nout, seed=303)
    res = rnd(1, M, Nout, seed=304)
    pw = ops.pack_linear_weight(w, b, device=DEV)
    xd, rd = bf(x).to(DEV), bf(res).to(DEV)
    y = ops.linear(xd, pw, residual=rd)
    assert_close(y, F.linear(x, w, b) + res, "p8 linear + residual, ragged M")
    for _ in range(60):
        assert torch.equal(ops.linear(xd, pw, residual=rd), y), "p8 linear: repeat differs"
    y32 = ops.linear(xd, pw, out_dtype=ops.OUT_F32)
    assert_close(y32, F.linear(x, w, b), "p8 f32 out", rel_l2=1e-5, max_ulps=0.05)
    inner = 256                                        # GEGLU: 512 packed columns
    wg = rnd(2 * inner, 2048, seed=305, scale=2048 ** -0.5)
    bg = rnd(2 * inner, seed=306, scale=0.1)
    xg = rnd(1, 25600, 2048, seed=307)
    h = F.linear(xg, wg, bg)
    a, g = h.chunk(2, dim=-1)
    yg = ops.linear(bf(xg).to(DEV), ops.pack_geglu_weight(wg, bg, device=DEV))
    assert_close(yg, a * F.gelu(g), "p8 geglu")
    B, Mb, Kb, Nb = 8, 2048, 1536, 1024                # 8 x 8 x 4 = 256 tiles through grid.z
    a_ = rnd(B, Mb, Kb, seed=308)
    b_ = rnd(B, Nb, Kb, seed=309)
    s = ops.bmm_nt(bf(a_).to(DEV), bf(b_).to(DEV), alpha=Kb ** -0.5, out_dtype=ops.OUT_F32)
    assert_close(s, torch.einsum("bmk,bnk->bmn", a_, b_) * Kb ** -0.5, "p8 bmm_nt", rel_l2=1e-5, max_ulps=0.05)


@pytest.mark.parametrize("M,K,Nout", [(36900, 1536, 1024), (20000, 3072, 2304), (66000, 1536, 256)])
def test_ping_pong_gemm_kernel_many_tiles(M, K, Nout):
    """igemm_p8_kernel with more tiles than CUs (several rounds of workgroups per CU, tile counts that do not divide by the XCD count,
    ragged M): bias + residual, f32 output, the same launch repeated. (Round 4 built a PERSISTENT form of this kernel - one workgroup
    per CU walking its XCD's tile range with the next tile's first K-tile fetched behind the current epilogue - and removed it: 2 %
    slower on F-1024 than letting the hardware dispatch one workgroup per tile, profiles/r04_experiments.md.)"""
    ops = _ops()
    x = rnd(1, M, K, seed=401)
    w = rnd(Nout, K, seed=402, scale=K ** -0.5)
    b = rnd(Nout, seed=403)
    res = rnd(1, M, Nout, seed=404)
    pw = ops.pack_linear_weight(w, b, device=DEV)
    xd, rd = bf(x).to(DEV), bf(res).to(DEV)
    assert ((M + 255) // 256) * ((Nout + 255) // 256) > 256
    y = ops.linear(xd, pw, residual=rd)
    assert_close(y, F.linear(x, w, b) + res, f"p8 linear + residual {M, K, Nout}")
    for _ in range(20):
        assert torch.equal(ops.linear(xd, pw, residual=rd), y), "p8: repeat differs"
    y32 = ops.linear(xd, pw, out_dtype=ops.OUT_F32)
    assert_close(y32, F.linear(x, w, b), "p8 f32 out", rel_l2=1e-5, max_ulps=0.05)


@pytest.mark.parametrize("kind,N,C,Cout,H,W,G", [
    ("halo-channel", 4, 320, 320, 64, 64, 32),        # UNet level 0: group size 10 -> one entry per channel, slot per wave tile
    ("halo-channel", 4, 320, 640, 32, 96, 32),        # group size 20
    ("dma-group", 4, 128, 128, 128, 96, 32),          # stride-2 conv on the GEMM kernel: slot per 32-row block, per group
    ("dma-channel", 4, 320, 320, 64, 64, 32),         # 1x1 conv (linear) with group size 10
    ("reg-group", 1, 64, 128, 32, 40, 32),            # small problem on the register-staged kernel
    ("linear", 6, 320, 320, 1, 1024, 32),             # ops.linear on [B, L, K]: B images of L rows
    ("reg-group", 2, 32, 32, 32, 32, 32),             # group size 1 (per channel == per group)
    ("reg-group", 2, 32, 64, 32, 32, 32),             # group size 2
    ("halo-channel", 8, 32, 128, 96, 64, 32),         # group size 4 through the halo path
])
def test_fused_groupnorm_statistics_all_paths(kind, N, C, Cout, H, W, G):
    """Every igemm path that can leave GroupNorm statistics (halo tile, GEMM-shaped row blocks; per group or per channel)
    against torch; the handle must be present, i.e. no read pass over the tensor."""
    ops = _ops()
    x = rnd(N, C, H, W, seed=300)
    gamma, beta = rnd(Cout, seed=304) + 1.0, rnd(Cout, seed=305)
    if kind.startswith("halo"):
        w = rnd(Cout, C, 3, 3, seed=301, scale=(9 * C) ** -0.5); b = rnd(Cout, seed=302)
        ref_conv = F.conv2d(x, w, b, padding=1)
        y = ops.conv2d(nhwc(x), ops.pack_conv_weight(w, b, device=DEV), pad=1, gn_groups=G)
    elif kind == "dma-group":
        w = rnd(Cout, C, 3, 3, seed=301, scale=(9 * C) ** -0.5); b = rnd(Cout, seed=302)
        ref_conv = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
        y = ops.conv2d(nhwc(x), ops.pack_conv_weight(w, b, device=DEV), stride=2, pad=(0, 1, 0, 1), gn_groups=G)
    elif kind == "reg-group":
        w = rnd(Cout, C, 3, 3, seed=301, scale=(9 * C) ** -0.5); b = rnd(Cout, seed=302)
        ref_conv = F.conv2d(x, w, b, padding=1)
        y = ops.conv2d(nhwc(x), ops.pack_conv_weight(w, b, device=DEV), pad=1, gn_groups=G)
    elif kind == "dma-channel":
        w = rnd(Cout, C, 1, 1, seed=301, scale=C ** -0.5); b = rnd(Cout, seed=302)
        r = rnd(N, Cout, H, W, seed=303)
        ref_conv = F.conv2d(x, w, b) + r
        y = ops.conv2d(nhwc(x), ops.pack_conv_weight(w, b, device=DEV), pad=0, residual=nhwc(r), gn_groups=G)
    else:
        w = rnd(Cout, C, seed=301, scale=C ** -0.5); b = rnd(Cout, seed=302)
        xt = x.view(N, C, H * W).transpose(1, 2).contiguous()                   # [B, L, K]
        ref_lin = F.linear(xt, w, b)                                             # [B, L, Cout]
        y = ops.linear(bf(xt).to(DEV), ops.pack_linear_weight(w, b, device=DEV), gn_groups=G)
        assert y.shape == ref_lin.shape
        ref_conv = ref_lin.transpose(1, 2).reshape(N, Cout, 1, H * W)
    assert getattr(y, "_omgsr_gn", None) is not None, f"{kind}: the producer should have left statistics"
    mean, rstd, var = ops.group_norm_stats(y, G, 1e-5)
    g = ref_conv.reshape(N, G, -1)
    assert torch.allclose(mean.cpu(), g.mean(-1), atol=3e-3, rtol=3e-3), kind
    assert torch.allclose(var.cpu(), g.var(-1, unbiased=False), atol=3e-3, rtol=6e-3), kind
    y4 = y if y.dim() == 4 else y.reshape(N, 1, H * W, Cout)
    out = ops.group_norm_apply(y4.contiguous(), mean, rstd, gamma.to(DEV), beta.to(DEV), G, ops.ACT_NONE)
    ref = F.group_norm(ref_conv, G, gamma, beta, eps=1e-5)
    assert_close(to_nchw(out), ref, f"{kind} groupnorm", rel_l2=6e-3, max_ulps=6.0)


@pytest.mark.parametrize("N,C,Cout,H,W,reps", [(3, 128, 128, 512, 512, 150), (3, 256, 256, 256, 256, 150), (36, 320, 320, 64, 64, 100)])
def test_conv_bit_repeatability(N, C, Cout, H, W, reps):
    """The same launch repeated must give the same BITS. This caught a real LDS race in the halo kernel: a fragment read
    still queued when the wave passed the K-step barrier could meet the next LDS-DMA write (one wave, a few weight rows,
    about one tile in 10^5) - every counted wait in front of a barrier now also retires the wave's LDS reads."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(N, H, W, C, generator=g) * 0.5).to(ops.act_dtype()).to(DEV)
    w = torch.randn(Cout, C, 3, 3, generator=g) * (9 * C) ** -0.5
    pw = ops.pack_conv_weight(w, torch.randn(Cout, generator=g), device=DEV)
    ref = ops.conv2d(x, pw, pad=1)
    bad = sum(0 if torch.equal(ops.conv2d(x, pw, pad=1), ref) else 1 for _ in range(reps))
    assert bad == 0, f"{bad} of {reps} repeats differ"


def test_linear_attention_norm_bit_repeatability():
    """Same property for the LDS-DMA GEMM (256x128, 256x256 and 192x128 tiles, split-K), attention, GroupNorm, LayerNorm."""
    ops = _ops()
    g = torch.Generator().manual_seed(78)
    dt = ops.act_dtype()

    def check(name, fn, reps):
        ref = fn()
        refs = ref if isinstance(ref, (tuple, list)) else (ref,)
        for _ in range(reps):
            cur = fn()
            curs = cur if isinstance(cur, (tuple, list)) else (cur,)
            assert all(torch.equal(a, b) for a, b in zip(refs, curs)), name

    for M, K, Nout in [(147456, 320, 320), (36864, 640, 1280), (9216, 1280, 1280), (2304, 2560, 1280), (4608, 3072, 3072)]:
        x = (torch.randn(1, M, K, generator=g) * 0.5).to(dt).to(DEV)
        pw = ops.pack_linear_weight(torch.randn(Nout, K, generator=g) * K ** -0.5, torch.randn(Nout, generator=g), device=DEV)
        res = (torch.randn(1, M, Nout, generator=g) * 0.5).to(dt).to(DEV)
        check(f"linear {M}x{K}->{Nout}", lambda: ops.linear(x, pw, residual=res), 40)
    B, L, H, D = 4, 4096, 5, 64
    qk = (torch.randn(B, L, 2 * H * D, generator=g) * 0.5).to(dt).to(DEV)
    vt = (torch.randn(B, H * D, L, generator=g) * 0.5).to(dt).to(DEV)
    check("attention", lambda: ops.attention(qk, qk, vt, H, D, D ** -0.5, q_col=0, k_col=H * D, Lk=L), 30)
    xg = (torch.randn(4, 256, 256, 256, generator=g)).to(dt).to(DEV)
    check("groupnorm stats", lambda: ops.group_norm_stats(xg, 32, 1e-6), 30)
    xl = (torch.randn(1, 36864, 640, generator=g)).to(dt).to(DEV)
    check("layernorm", lambda: ops.layer_norm(xl, None, None, 1e-5), 30)


# ---- GroupNorm apply (+ SiLU) as the conv's patch producer (round 5: omgsr_igemm_args.gn_scale_shift) ---------------
# N rows, images (rows share the statistics of image n % nimg), C, Cout, H, W, act (1 = SiLU), residual, fusable expected
GN_CONV_CASES = [
    (12, 4, 128, 128, 64, 64, 1, True, True),       # (the halo-tile kernel wants >= 192 workgroup tiles: 16 per 64 x 64 image)
    (12, 2, 128, 128, 43, 86, 1, False, True),      # tile-major rows sharing two images' statistics, ragged map
    (8, 2, 256, 128, 64, 96, 1, True, True),
    (4, 2, 512, 512, 48, 64, 1, False, False),      # four column tiles: the shipped policy keeps the apply pass (OMGSR_GN_FUSE_MAX_COUT)
    (8, 2, 512, 128, 96, 64, 1, False, True),
    (12, 1, 1024, 128, 64, 64, 1, False, True),     # the largest (scale, shift) table the kernel holds (8 KB)
    (2, 2, 128, 3, 128, 192, 1, False, True),       # conv_out: the narrow shape
    (12, 4, 128, 128, 64, 64, 0, False, False),     # no activation: SiLU is the one the producer applies -> apply pass
    (4, 1, 64, 128, 96, 160, 1, True, True),        # two K chunks only: the first one is normalised in the prologue
    (4, 2, 32, 128, 96, 160, 1, False, True),       # ONE chunk: prologue only
    (1, 1, 128, 128, 16, 16, 1, False, False),      # too few tiles for the halo kernel: the apply pass runs in front of the conv
    (30, 2, 128, 256, 38, 38, 1, True, False),      # narrow map: FLAT form, no normalising instantiation -> apply pass
    (12, 1, 2048, 128, 64, 64, 1, False, False),    # table too large -> apply pass
]


def _gn_ref(x, mean, rstd, gamma, beta, G, act, nimg):
    N, Cc = x.shape[0], x.shape[1]
    idx = torch.arange(N) % nimg
    m = mean[idx].repeat_interleave(Cc // G, 1)[:, :, None, None]
    r = rstd[idx].repeat_interleave(Cc // G, 1)[:, :, None, None]
    y = (x - m) * r * gamma[None, :, None, None] + beta[None, :, None, None]
    return F.silu(y) if act else y


@pytest.mark.parametrize("case", GN_CONV_CASES)
def test_conv3x3_groupnorm_fused_into_the_patch_producer(case):
    """conv2d(x, gn=spec) == conv2d(group_norm_apply(x)) (the same values: one fp32 (scale, shift) table, one 16-bit rounding of the
    normalised operand) and == the fp32 torch reference within the 16-bit operand's rounding; zero padding applies to the NORMALISED map."""
    ops = _ops()
    from omgsr_amd import _lib
    import ctypes as C
    N, nimg, Cc, Cout, H, W, act, use_res, expect_fused = case
    G = 32
    x = rnd(N, Cc, H, W, seed=11) * 1.5 + 0.25
    w = rnd(Cout, Cc, 3, 3, seed=12, scale=1.0 / math.sqrt(9 * Cc))
    b = rnd(Cout, seed=13)
    gamma, beta = 1.0 + 0.2 * rnd(Cc, seed=14), 0.3 * rnd(Cc, seed=15)
    mean, rstd = 0.25 + 0.1 * rnd(nimg, G, seed=16), (1.0 / 1.5) * (1.0 + 0.1 * rnd(nimg, G, seed=17)).abs()
    # the normalised tensor is an MFMA operand: rounded ONCE to the compute type (by the apply pass and by the fused producer alike)
    xn = _gn_ref(x, mean, rstd, gamma, beta, G, act, nimg).to(ops.act_dtype()).float()
    ref = F.conv2d(F.pad(xn, (1, 1, 1, 1)), w, b)
    res = None
    if use_res:
        res = rnd(*ref.shape, seed=18)
        ref = ref + res
    pw = ops.pack_conv_weight(w, b, device=DEV)
    xd = nhwc(x)
    spec = ops.GnSpec(mean.to(DEV), rstd.to(DEV), gamma.to(DEV), beta.to(DEV), G, ops.ACT_SILU if act else ops.ACT_NONE)
    # what the library says about this problem
    a = _lib.IgemmArgs()
    ops._conv_args(a, xd, pw, 1, 1, False, ops.ACT_NONE, None, None, ops.OUT_STREAM, 1.0, None, 1, 0)
    a.in_el = ops.EL_16
    assert ops._gn_fusable(a, spec) == expect_fused
    rd = None if res is None else nhwc(res)
    y = ops.conv2d(xd, pw, pad=1, residual=rd, gn=spec, gn_groups=32 if Cout >= 96 else 0)
    y_unfused = ops.conv2d(spec.apply(xd), pw, pad=1, residual=rd)
    # (looser than the plain conv tests: the GPU evaluates the affine as one fma from the fp32 table, the reference as (x - m) r g + b, so a few
    # operand elements round to the neighbouring 16-bit value - each flip is one operand ulp in a sum of 9 C terms)
    assert_close(to_nchw(y)[:, :Cout], ref, "gn-fused conv vs fp32 reference", rel_l2=6e-3, max_ulps=6.0)
    # against the two-pass form: identical operand bits are expected (same table, same expression); allow one 16-bit ulp of the OUTPUT
    # for a contraction the compiler may have fused differently in the two kernels
    d = (y.float() - y_unfused.float()).abs().max().item()
    assert d <= 2 ** -7 * max(1.0, float(ref.abs().max())), f"fused vs apply + conv differ by {d}"
    if Cout >= 96 and expect_fused:                  # the fused conv still leaves ITS output's GroupNorm statistics
        m2, r2, _ = ops.group_norm_stats(y, 32, 1e-6)
        yr = y.float().cpu().permute(0, 3, 1, 2)
        mr = yr.reshape(N, 32, -1).mean(-1)
        assert torch.allclose(m2.cpu(), mr, atol=2e-3, rtol=1e-3)


def test_conv3x3_groupnorm_fused_multi_launch():
    """The tile-shape groups of one tiled-VAE layer through conv2d_multi(gn=...): one launch, every group normalised with the statistics of
    its rows' images (row r of a tile-major group belongs to image r % nimg)."""
    ops = _ops()
    Cc, Cout, G, nimg = 128, 128, 32, 2
    w = rnd(Cout, Cc, 3, 3, seed=21, scale=1.0 / math.sqrt(9 * Cc))
    b = rnd(Cout, seed=22)
    gamma, beta = 1.0 + 0.2 * rnd(Cc, seed=23), 0.3 * rnd(Cc, seed=24)
    mean, rstd = 0.1 * rnd(nimg, G, seed=25), (1.0 + 0.1 * rnd(nimg, G, seed=26)).abs()
    shapes = [(8, 86, 86), (4, 86, 64), (4, 64, 86), (2, 64, 64)]
    xs = [rnd(n, Cc, h, ww, seed=30 + i) for i, (n, h, ww) in enumerate(shapes)]
    rs = [rnd(n, Cout, h, ww, seed=40 + i) for i, (n, h, ww) in enumerate(shapes)]
    pw = ops.pack_conv_weight(w, b, device=DEV)
    spec = ops.GnSpec(mean.to(DEV), rstd.to(DEV), gamma.to(DEV), beta.to(DEV), G, ops.ACT_SILU)
    ys = ops.conv2d_multi([nhwc(x) for x in xs], pw, pad=1, residuals=[nhwc(r) for r in rs], gn=spec, gn_groups=32)
    for x, r, y in zip(xs, rs, ys):
        ref = F.conv2d(F.pad(_gn_ref(x, mean, rstd, gamma, beta, G, 1, nimg).to(ops.act_dtype()).float(), (1, 1, 1, 1)), w, b) + r
        assert_close(to_nchw(y), ref, "gn-fused multi conv", rel_l2=6e-3, max_ulps=6.0)
        assert getattr(y, "_omgsr_gn", None) is not None


# ---- round 5: the load-time constant folds in the library's own fp32 kernel (no vendor BLAS in the product process) ---------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("rows,K,N,silu,bias", [(1, 320, 1280, False, True), (1, 1280, 1280, True, True), (1, 3072, 18432, True, True),
                                                (1, 768, 3072, False, True), (3, 257, 129, True, False), (3072, 16, 3072, False, False)])
def test_linear_f32_constant_fold(rows, K, N, silu, bias):
    """omgsr_linear_f32 against fp64 math: time / guidance / text embeddings, adaLN modulation, time_emb_proj and the LoRA merge's rank-r
    product (rows = out, K = r). fp32 FMAs in a fixed order: repeat launches give the same bits."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows * 7 + K)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g) if bias else None
    ref = F.linear(F.silu(x.double()) if silu else x.double(), w.double(), None if b is None else b.double())
    xd, wd, bd = x.cuda(), w.cuda(), None if b is None else b.cuda()
    got = ops.linear_f32(xd, wd, bd, silu_in=silu)
    assert got.dtype == torch.float32 and got.shape == (rows, N)
    err = (got.double().cpu() - ref).abs().max().item()
    assert err < 2e-6 * max(1.0, ref.abs().max().item()) * max(1.0, (K / 64) ** 0.5), err
    assert torch.equal(got, ops.linear_f32(xd, wd, bd, silu_in=silu))
    # 16-bit checkpoints: the weights widen to fp32 first, exactly like the reference's `.float()` folds
    got16 = ops.linear_f32(xd, wd.bfloat16(), bd, silu_in=silu)
    ref16 = F.linear(F.silu(x.double()) if silu else x.double(), w.bfloat16().double(), None if b is None else b.double())
    assert (got16.double().cpu() - ref16).abs().max().item() < 2e-6 * max(1.0, ref16.abs().max().item()) * max(1.0, (K / 64) ** 0.5)

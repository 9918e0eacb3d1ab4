"""Host-side logic of the accurate tier on CPU (no kernels run): the shipped precision policy's layer assignment, its consistency
checks and fingerprint, the weight packings of the split forms (two-term K-concatenation with the wrapped contraction, the
mixed-precision fp16 + fp8 form, the phase-summed kernels of the upsampling convs)."""
import pytest
import torch
import torch.nn.functional as F


@pytest.fixture()
def accurate_tier():
    from omgsr_amd import ops
    ops.set_compute_dtype(torch.float32)          # host-side switch only (omgsr_set_compute_dtype touches no device)
    yield ops
    ops.set_compute_dtype(torch.bfloat16)


def _forms(model, kinds=None):
    from omgsr_amd import nn as N
    out = {}
    for _, m in model.named_modules():
        if isinstance(m, kinds or (N.Conv2d, N.Linear)):
            out[(m.op_split, m.w_split)] = out.get((m.op_split, m.w_split), 0) + 1
    return out


def test_shipped_policy_assignment_on_sd21_shapes():
    from omgsr_amd import precision as P
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    with torch.device("meta"):
        v, u = AutoencoderKL(), UNet2DConditionModel()
    P.apply_default_policy(vae=v, unet=u)
    fv, fu = _forms(v), _forms(u)
    # VAE: 30 resnet convs + 3 upsampling convs + (round 4) the 4 1x1 shortcut convs in the mixed-precision form, the decoder's 18 resnet
    # convs above 64 px single, everything else (samplers, conv_in / out, quant convs and - since the 40 x 2-draw sweep of round 4 - the 8
    # mid-block attention linears) split on both sides
    # round 5: the 3x3 convs of the halo-tile kernel (resnet convs: operand from a GroupNorm apply; the VAE's up-samplers: operand from the previous
    # conv's OUT6 epilogue) carry their correction segments as fp6 with per-block scales (op_split 4); the 1x1 shortcuts stay fp8
    assert fv == {(4, 2): 33, (3, 2): 4, (1, 1): 18, (2, 2): 17}
    assert [n for n, m in v.named_modules() if getattr(m, "qk_split", False)] == ["decoder.mid_block.attentions.0"]     # q / k of the decoder's attention split
    mx = [n for n, m in v.named_modules() if getattr(m, "op_split", 0) in (3, 4)]
    assert all(m.kernel_size == (3, 3) for n, m in v.named_modules() if getattr(m, "op_split", 0) == 4)
    assert all(m.kernel_size == (1, 1) for n, m in v.named_modules() if getattr(m, "op_split", 0) == 3)
    assert all(("resnets" in n or "upsamplers" in n) for n in mx) and not any(("decoder.up_blocks.1.resnets" in n and "shortcut" not in n) for n in mx)
    # UNet convs: the 64 x 64 and 32 x 32 resnet convs + the three upsampling convs + every 1x1 shortcut in the mixed-precision form; 16 x 16 resnets and
    # the 8 x 8 level + mid block single. UNet linears (round 4): the 64 x 64 / 32 x 32 transformer blocks' both-sides splits and every
    # proj_in / proj_out run as mixed-precision GEMMs (igemm_gmx_kernel); the 16 x 16 transformer blocks are single (round-4 trim);
    # 64 x 64 q / k / v weight-split only; the 32 x 32 cross-attention K / V (the prompt's projections) keep the two-term split
    from omgsr_amd import nn as N
    assert _forms(u, N.Conv2d) == {(2, 2): 5, (4, 2): 20, (3, 2): 17, (1, 1): 24}            # (4, 2): 20 resnet convs (fp6); (3, 2): 3 upsampling convs + 14 1x1 shortcuts
    assert _forms(u, N.Linear) == {(1, 2): 49, (2, 2): 15, (3, 2): 92, (1, 1): 60} and sum(fu.values()) == 282
    assert u.down_blocks[2].resnets[0].conv1.w_split == 1 and u.down_blocks[0].attentions[0].transformer_blocks[0].attn1.to_q.w_split == 2
    b32, b16 = u.down_blocks[1].attentions[0].transformer_blocks[0], u.down_blocks[2].attentions[0].transformer_blocks[0]
    assert [m.op_split for m in (b32.attn1.to_q, b32.attn1.to_k, b32.attn1.to_v, b32.attn1.to_out[0], b32.attn2.to_q, b32.ff.net[0].proj, b32.ff.net[2])] == [3] * 7
    assert (b32.attn2.to_k.op_split, b32.attn2.to_v.op_split) == (2, 2)
    assert {(m.op_split, m.w_split) for m in b16.modules() if isinstance(m, N.Linear)} == {(1, 1)}
    assert u.down_blocks[2].attentions[0].proj_in.op_split == 3 and u.down_blocks[0].attentions[0].transformer_blocks[0].attn1.to_q.op_split == 1
    a, b = P.policy_fingerprint(u), P.policy_fingerprint(v)
    P.set_weight_split(u, [])
    assert P.policy_fingerprint(u) != a and P.policy_fingerprint(v) == b
    P.resolve("all", unet=u)
    assert _forms(u) == {(2, 2): 282}


def test_check_policy_rejects_layers_that_share_an_operand_but_disagree():
    from omgsr_amd import precision as P
    from omgsr_amd.diffusers_api import FluxTransformer2DModel, UNet2DConditionModel
    with torch.device("meta"):
        u = UNet2DConditionModel(block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4], cross_attention_dim=128)
        f = FluxTransformer2DModel(num_layers=1, num_single_layers=1, num_attention_heads=2, attention_head_dim=128, joint_attention_dim=64,
                                   pooled_projection_dim=32, in_channels=64)
    with pytest.raises(ValueError, match="to_q / to_k / to_v"):
        P.set_operand_split(u, [r"attn1\.to_q$"])
    with pytest.raises(ValueError, match="one fused weight"):
        P.set_weight_split(u, [r"attn1\.to_q$"])
    with pytest.raises(ValueError, match="add_q_proj / add_k_proj / add_v_proj"):
        P.set_operand_split(f, [r"add_q_proj$"])
    with pytest.raises(ValueError, match="to_out.0 / to_add_out"):
        P.set_operand_split(f, [r"to_add_out$"])
    P.set_operand_split(f, [r"to_add_out$", r"attn\.to_out\.0$"])       # together: fine


def test_split_weight_packings(accurate_tier):
    ops = accurate_tier
    g = torch.Generator().manual_seed(0)
    C, Cout = 64, 16
    w = torch.randn(Cout, C, 3, 3, generator=g) * 0.05
    w_hi = w.to(torch.float16).float()
    # K-concatenation per tap: [w_hi] * split, then [w_lo]
    pw = ops.pack_conv_weight(w, None, device="cpu", split=2, w_split=2)
    assert pw.cin == 3 * C and pw.row_channels == 2 * C and pw.k_pad == 9 * 3 * C
    rows = pw.w[:Cout].float().reshape(Cout, 9, 3 * C)
    ref = w.permute(0, 2, 3, 1).reshape(Cout, 9, C)
    assert torch.equal(rows[..., :C], w_hi.permute(0, 2, 3, 1).reshape(Cout, 9, C)) and torch.equal(rows[..., C:2 * C], rows[..., :C])
    assert torch.equal(rows[..., 2 * C:], (ref - rows[..., :C]).to(torch.float16).float())
    assert ((rows[..., :C] + rows[..., 2 * C:]) - ref).abs().max() < 2.0 ** -20 * ref.abs().max()
    pw = ops.pack_conv_weight(w, None, device="cpu", split=1, w_split=2)
    assert pw.cin == 2 * C and pw.row_channels == C           # the contraction wraps over the operand row (omgsr_igemm_args.in_ld)
    # weights that are exact in the compute type have w_lo == 0: the low segment is dropped at pack time (same bits, less work)
    pw = ops.pack_conv_weight(w_hi, None, device="cpu", split=2, w_split=2)
    assert pw.w_split == 1 and pw.cin == 2 * C and pw.row_channels == 2 * C and torch.equal(pw.w[:Cout].float().reshape(Cout, 9, 2 * C), rows[..., :2 * C])
    assert ops.pack_linear_weight(w_hi[:, :, 0, 0], None, device="cpu", split=1, w_split=2).cin == C
    # the mixed-precision form: per tap 4C bytes [w_hi fp16 | fp8(w_hi 2^s1) | fp8(w_lo 2^s2)], scales as E8M0 exponents
    pw = ops.pack_conv_weight(w, None, device="cpu", split=3)
    n16, e_w1, e_a1, e_w2, e_a2 = pw.mx
    assert n16 == C // 32 and e_a1 == 127 - 11 and e_a2 == 127 and pw.cin == 2 * C and pw.row_channels == 2 * C
    by = pw.w[:Cout].view(torch.uint8).reshape(Cout, 9, 4 * C)
    hi = by[..., :2 * C].contiguous().view(torch.float16).float()
    assert torch.equal(hi, rows[..., :C])
    hi8 = by[..., 2 * C:3 * C].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** (e_w1 - 127)
    lo8 = by[..., 3 * C:].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** (e_w2 - 127)
    assert (hi8 - hi).abs().max() <= 2.0 ** -4 * hi.abs().max() and (lo8 - (ref - hi)).abs().max() <= 2.0 ** -4 * (ref - hi).abs().max()
    assert ((hi + lo8) - ref).abs().max() < 2.0 ** -14 * ref.abs().max()           # w_hi + w_lo' carries w to ~2^-15
    with pytest.raises(ValueError, match="Cin % 64"):
        ops.pack_conv_weight(torch.randn(16, 32, 3, 3), None, device="cpu", split=3)
    # ... and of a Linear (round 4: igemm_gmx_kernel streams the plain row-major packing, there is no slice-major copy)
    wl = torch.randn(200, 320, generator=g) * 320 ** -0.5
    pl = ops.pack_linear_weight(wl, torch.zeros(200), device="cpu", split=3)
    assert pl.mx is not None and pl.w_cm is None and (pl.R, pl.S) == (1, 1) and pl.cin == 640 and pl.k_pad == 640 and pl.cout_pad == 256 and pl.mx[0] == 10
    byl = pl.w[:200].view(torch.uint8).reshape(200, 4 * 320)
    hl = byl[:, :640].contiguous().view(torch.float16).float()
    assert torch.equal(hl, wl.to(torch.float16).float()) and not pl.w[200:].any()
    l8 = byl[:, 960:].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** (pl.mx[3] - 127)
    assert ((hl + l8) - wl).abs().max() < 2.0 ** -14 * wl.abs().max()
    pgl = ops.pack_geglu_weight(torch.randn(2 * 128, 64, generator=g), torch.randn(2 * 128, generator=g), device="cpu", split=3)
    assert pgl.geglu and pgl.cout == 128 and pgl.mx is not None and pgl.cin == 128


_E2M3_GRID = [0, .125, .25, .375, .5, .625, .75, .875, 1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875, 2, 2.25, 2.5, 2.75, 3, 3.25, 3.5, 3.75, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5]


def _decode_mx6_third(raw):
    """One correction third of an OMGSR_EL_MX6 row, straight from the format's definition (include/omgsr_hip.h)."""
    grid = torch.tensor(_E2M3_GRID, dtype=torch.float64)
    lead, C = raw.shape[:-1], raw.shape[-1]
    g = raw.reshape(-1, C // 64, 64).to(torch.int64)
    out = torch.zeros(g.shape[0], C // 64, 2, 32, dtype=torch.float64)
    for h in (0, 1):
        st = torch.cat([g[..., 16 * h:16 * h + 16], g[..., 32 + 16 * h:40 + 16 * h]], -1)
        sc = 2.0 ** (g[..., 40 + 16 * h].double() - 127)
        for i in range(32):
            by, sh = (6 * i) // 8, (6 * i) % 8
            code = ((st[..., by] | (st[..., min(by + 1, 23)] << 8)) >> sh) & 63
            out[..., h, i] = torch.where((code & 32) != 0, -1.0, 1.0) * grid[code & 31] * sc
    return out.reshape(*lead, C)


def test_fp6_weight_packing(accurate_tier):
    """split 4 (OMGSR_EL_MX6): [w_hi fp16 | e2m3 blocks of w_hi | e2m3 blocks of w - w_hi] per tap; every 32-channel block carries its own E8M0
    scale, codes round to nearest (ties to even) and saturate at 7.5, padding bytes are zero; 3x3 stride-1 nine-tap convs only."""
    ops = accurate_tier
    g = torch.Generator().manual_seed(5)
    C, Cout = 128, 16
    w = torch.randn(Cout, C, 3, 3, generator=g) * 0.05 * torch.exp(torch.randn(C, 1, 1, generator=g))
    pw = ops.pack_conv_weight(w, None, device="cpu", split=4)
    assert pw.split == 4 and pw.mx_fmt == 6 and pw.mx[0] == C // 32 and pw.cin == 2 * C and pw.row_channels == 2 * C and pw.w_ph is None
    assert pw.w_cm is not None and tuple(pw.w_cm.shape) == (2 * C // 32, 9, pw.cout_pad, 32)
    by = pw.w[:Cout].view(torch.uint8).reshape(Cout, 9, 4 * C)
    ref = w.permute(0, 2, 3, 1).reshape(Cout, 9, C)
    hi = by[..., :2 * C].contiguous().view(torch.float16).float()
    assert torch.equal(hi, ref.to(torch.float16).float())
    hi6, lo6 = _decode_mx6_third(by[..., 2 * C:3 * C]), _decode_mx6_third(by[..., 3 * C:])
    bmax = lambda t: t.reshape(Cout, 9, C // 32, 32).abs().amax(-1, keepdim=True).expand(Cout, 9, C // 32, 32).reshape(Cout, 9, C)      # noqa: E731
    lo = (ref - hi).double()
    assert ((hi6 - hi.double()).abs() <= bmax(hi.double()) * (0.25 / 3.75)).all() and ((lo6 - lo).abs() <= bmax(lo) * (0.25 / 3.75)).all()
    assert ((hi.double() + lo6) - ref.double()).abs().max() < 2.0 ** -14 * ref.abs().max()
    raw = by[..., 2 * C:].reshape(Cout, 9, 2 * C // 64, 64)
    assert not raw[..., 41:48].any() and not raw[..., 57:64].any()                 # the 7 + 7 padding bytes of a 64-byte group
    # exact grid points survive, ties go to even codes, anything beyond 7.5 x the block scale saturates
    v = torch.zeros(1, 64)
    v[0, :8] = torch.tensor([4.0, 4.25, 4.75, 7.5, 7.9, -0.0625, 0.1875, -3.0])    # block max 7.9 -> scale 2^0 (top binade [4, 8))
    d = _decode_mx6_third(ops._e2m3_blocks(v))[0, :8]
    assert d.tolist() == [4.0, 4.0, 5.0, 7.5, 7.5, -0.0, 0.25, -3.0]
    assert ops._e2m3_blocks(torch.zeros(2, 64)).sum() == 0                          # an all-zero block: scale byte 0, codes 0
    with pytest.raises(ValueError, match="3x3"):
        ops.pack_conv_weight(torch.randn(16, 64, 1, 1), None, device="cpu", split=4)
    pu = ops.pack_conv_weight(w, None, device="cpu", split=4, upsample_phases=True)          # the up-samplers' phase-summed 2 x 2 kernels in the same form
    assert pu.w_ph is not None and tuple(pu.w_ph.shape) == (4, 2 * C // 32, 4, pu.cout_pad, 32) and pu.mx_fmt == 6


def test_phase_summed_kernels_equal_the_upsampled_conv():
    """conv3x3(nearest_up2(x)) at output pixel (2y + a, 2x + b) == the 2 x 2 convolution of x with the phase-(a, b) summed taps."""
    from omgsr_amd.ops import _phase_kernels
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(2, 5, 7, 6, generator=g).double(), torch.randn(4, 5, 3, 3, generator=g).double()
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    ph = _phase_kernels(w.permute(0, 2, 3, 1).contiguous().float()).double()          # [4, Cout, 2, 2, C]
    out = torch.zeros_like(ref)
    for a in (0, 1):
        for b in (0, 1):
            out[:, :, a::2, b::2] = F.conv2d(F.pad(x, (1 - b, b, 1 - a, a)), ph[2 * a + b].permute(0, 3, 1, 2))
    assert (out - ref).abs().max() < 1e-5


def test_w_lo_segment_of_a_phase_form_conv_is_kept_when_the_phase_sums_are_not_exact():
    """Round 6: pack_conv_weight drops the [w_lo] K segment when every weight is exact in the compute type. An up-sampling conv also carries its four
    phase-summed 2 x 2 kernels (sums of 2 - 4 taps): bf16-representable TAPS do not make their SUMS bf16-representable, so in the range-fallback tier
    (bf16 operands) the sums were rounded to 8 bits with no w_lo to carry the rest - 3.4e-3 against the oracle where full-mantissa weights measure
    1.1e-3. The segment now goes only if the phase sums are exact too; both packings of one layer keep the same K layout."""
    from omgsr_amd import ops
    ops.set_compute_dtype(torch.float32, operand_dtype=torch.bfloat16)       # the range-fallback tier's operand type (host-side switch)
    try:
        g = torch.Generator().manual_seed(3)
        w = (torch.randn(64, 32, 3, 3, generator=g) * 0.05).to(torch.bfloat16).float()          # exact in bf16
        plain = ops.pack_conv_weight(w, None, device="cpu", split=2, w_split=2)
        assert plain.w_split == 1                                              # nothing to carry: the segment is dropped (same bits, less work)
        up = ops.pack_conv_weight(w, None, device="cpu", split=2, w_split=2, upsample_phases=True)
        ph = ops._phase_kernels(w.permute(0, 2, 3, 1))
        assert not torch.equal(ph.to(torch.bfloat16).float(), ph)              # the sums really are inexact in bf16
        assert up.w_split == 2 and up.w_ph is not None
        # [4][Cin/32][4 taps][Cout_pad][32] with Cin = 32 * 3 segments: hi | hi | lo of the phase sums reproduce them to 2^-16
        seg = up.w_ph.float().reshape(4, 3, 4, up.w_ph.shape[3], 32)[:, :, :, :64]                # [phase][segment][tap][cout][32]
        want = ph.reshape(4, 64, 4, 32).permute(0, 2, 1, 3)                                       # [phase][tap][cout][c]
        assert torch.equal(seg[:, 0], seg[:, 1])
        assert ((seg[:, 0] + seg[:, 2]).permute(0, 1, 2, 3) - want).abs().max() <= want.abs().max() * 2.0 ** -15
        # exact sums (a 1-tap-per-phase kernel): dropped again
        w1 = torch.zeros(64, 32, 3, 3); w1[:, :, 1, 1] = w[:, :, 1, 1]
        assert ops.pack_conv_weight(w1, None, device="cpu", split=2, w_split=2, upsample_phases=True).w_split == 1
    finally:
        ops.set_compute_dtype(torch.bfloat16)


def test_attn_split_follows_the_tier(monkeypatch):
    """ops.attn_split(): the two-term split q / k / P / V inside the attention kernels belongs to the range-fallback tier (fp32 stream, bf16 operands);
    OMGSR_ATTN_SPLIT forces it off / on in the accurate tiers only (A/B runs), never in the fast tiers."""
    from omgsr_amd import ops
    try:
        monkeypatch.delenv("OMGSR_ATTN_SPLIT", raising=False)
        ops.set_compute_dtype(torch.bfloat16); assert not ops.attn_split()
        ops.set_compute_dtype(torch.float32); assert not ops.attn_split()
        ops.set_compute_dtype(torch.float32, operand_dtype=torch.bfloat16); assert ops.attn_split()
        monkeypatch.setenv("OMGSR_ATTN_SPLIT", "0"); assert not ops.attn_split()
        monkeypatch.setenv("OMGSR_ATTN_SPLIT", "1")
        ops.set_compute_dtype(torch.float32); assert ops.attn_split()
        ops.set_compute_dtype(torch.bfloat16); assert not ops.attn_split()
    finally:
        ops.set_compute_dtype(torch.bfloat16)

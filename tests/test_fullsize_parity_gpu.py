"""North-star parity at FULL sizes (BASELINE.json: PSNR >= 60 dB and rel-L2 <= 1e-3 against the fp32 reference path): the
accurate tier (`--weight_dtype fp32`) of the HIP pipeline against the fp32 CPU oracle with identical seeded weights, inputs
and posterior noise, at the real SD2.1-base / FLUX.1-dev layer shapes:

  * OMGSR-S 128->512, batch 1                          (BASELINE configs[0]/[1]; oracle ~6 s on the GPU box's host)
  * OMGSR-S 256->1024, tiled VAE enc 256 / dec 64, batch 4, images 0 AND 3 compared   (configs[2]; oracle ~35 s per image)
  * FluxTransformer2DModel at full WIDTH (D = 3072, 24 heads x 128, 4096 image + 512 text tokens, joint_attention_dim 4096)
    and reduced DEPTH (2 double + 2 single blocks: the fp32 oracle of all 57 blocks needs 48 GB and ~90 TFLOP on the CPU)
The fast tiers (bf16 = the reference's default dtype, fp16) run the same comparison against THEIR bound, which is what
16-bit activation storage allows (tests/emulate_numerics.py reproduces both numbers on the CPU), not the north-star's."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

NORTH_STAR_REL_L2, NORTH_STAR_PSNR = 1e-3, 60.0


@pytest.fixture(scope="module")
def s_oracle():
    """fp32 CPU oracle outputs of the two OMGSR-S configurations, computed once for every tier."""
    from omgsr_amd.testing import seeded_init_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef, TiledVaeRef
    torch.set_num_threads(min(16, torch.get_num_threads()))
    vae, unet = seeded_init_(R.AutoencoderKL(), 101).eval(), seeded_init_(R.UNet2DConditionModel(), 202).eval()
    prompt = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(4321)).to(torch.bfloat16).float()
    alpha = R.DDPMScheduler().alphas_cumprod[273]
    out = {"prompt": prompt, "sd": (vae.state_dict(), unet.state_dict())}
    with torch.no_grad():
        x = synthetic_lq(1, 512, 512, seed=1234)
        eps = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(99))
        vae.posterior_noise = eps
        out["s512"] = (x, eps, OmgsrSRef(vae, unet, alpha, 273)(x, prompt, 64, 32))
        x = synthetic_lq(4, 1024, 1024, seed=1234)
        eps = torch.randn(4, 4, 128, 128, generator=torch.Generator().manual_seed(99))
        # both ends of the batch of 4: images 0 and 3, one oracle run each (~35 s)
        refs = []
        for i in (0, 3):
            vae.posterior_noise = eps[i:i + 1]
            refs.append(OmgsrSRef(TiledVaeRef(vae, 256, 64), unet, alpha, 273)(x[i:i + 1], prompt, 64, 32))
        out["s1024t"] = (x, eps, torch.cat(refs, 0))
    return out


def _pipe(s_oracle, dtype):
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    vae, unet = AutoencoderKL(), UNet2DConditionModel()
    vae.load_state_dict(s_oracle["sd"][0]); unet.load_state_dict(s_oracle["sd"][1])
    return OMGSR_S_Infer(None, None, 273, DEV, dtype, vae=vae, unet=unet)


TIERS = [(torch.float32, NORTH_STAR_REL_L2, NORTH_STAR_PSNR), (torch.float16, 3e-3, 60.0), (torch.bfloat16, 2.5e-2, 43.0)]


@pytest.mark.parametrize("dtype,tol,min_psnr", TIERS, ids=["fp32-accurate", "fp16-fast", "bf16-fast"])
@pytest.mark.parametrize("config", ["s512", "s1024t"])
def test_omgsr_s_full_size_vs_oracle(s_oracle, config, dtype, tol, min_psnr):
    from omgsr_amd import ops
    from omgsr_amd.testing import psnr, rel_l2
    x, eps, ref = s_oracle[config]
    try:
        pipe = _pipe(s_oracle, dtype)
        if config == "s1024t":
            pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
        pipe.vae.posterior_noise = eps.to(DEV)
        with torch.no_grad():
            got, _ = pipe(x.to(DEV), s_oracle["prompt"].to(DEV), 64, 32)
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    assert torch.isfinite(got).all()
    which = (0, 3) if config == "s1024t" else (0,)
    for j, i in enumerate(which):
        gi, ri = got[i:i + 1].float().cpu(), ref[j:j + 1]
        e, p = rel_l2(gi, ri), psnr(gi, ri)
        print(f"OMGSR-S {config} {dtype} image {i}: rel-L2 {e:.3e}  PSNR {p:.1f} dB (bound {tol:g} / {min_psnr} dB)")
        assert gi.shape == ri.shape
        assert e <= tol and p >= min_psnr


def test_flux_full_width_vs_oracle():
    """FLUX.1-dev layer shapes (D 3072, 24 x 128 heads, 4096 + 512 tokens), 2 + 2 blocks, accurate tier vs the fp32 oracle."""
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import FluxTransformer2DModel
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import prepare_latent_image_ids
    cfg = dict(num_layers=2, num_single_layers=2)
    o = seeded_init_(R.FluxTransformer2DModel(**cfg), 404).eval()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 4096, 64, generator=g)
    pe, pooled = torch.randn(1, 512, 4096, generator=g), torch.randn(1, 768, generator=g)
    tids, iids = torch.zeros(512, 3), prepare_latent_image_ids(64, 64)
    t, gd = torch.tensor([0.5051124691963196]), torch.full((1,), 1.0)
    with torch.no_grad():
        ref = o(hidden_states=x, timestep=t, guidance=gd, pooled_projections=pooled, encoder_hidden_states=pe, txt_ids=tids,
                img_ids=iids, return_dict=False)[0]
    try:
        ops.set_compute_dtype(torch.float32)
        p = FluxTransformer2DModel(**cfg)
        p.load_state_dict(o.state_dict())
        del o
        p = p.to(DEV, torch.float32).eval()
        from omgsr_amd.precision import apply_default_policy
        apply_default_policy(flux=p)
        with torch.no_grad():
            got = p(hidden_states=x.to(DEV), timestep=t.to(DEV), guidance=gd.to(DEV), pooled_projections=pooled.to(DEV),
                    encoder_hidden_states=pe.to(DEV), txt_ids=tids.to(DEV), img_ids=iids.to(DEV), return_dict=False)[0]
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    e = rel_l2(got, ref)
    print(f"Flux full width 2+2 blocks, accurate tier: rel-L2 {e:.3e}")
    assert got.dtype == torch.float32 and torch.isfinite(got).all()
    assert e <= NORTH_STAR_REL_L2


# ---- robustness of the accurate tier's claim ---------------------------------------------------------------
ROBUST_REL_L2 = 8e-4          # 25 % head-room under the north-star's 1e-3


# OMGSR_ROBUST_DRAWS=<n> widens the sweep beyond the three weight draws the suite runs by default (DESIGN.md §4 records a 40-draw run).
# Draw 16 is the sentinel of that sweep: 1.26e-3 on its input draw 0 under the policies of rounds 3 and 4 until the VAE's mid-attention
# operands (and the decoder's q / k) were split (omgsr_amd/precision.py), 6.2e-4 since - the worst of the 80 cases; it always runs.
_NDRAWS = int(os.environ.get("OMGSR_ROBUST_DRAWS", "3"))
@pytest.mark.parametrize("wseed", list(range(_NDRAWS)) + ([16] if _NDRAWS <= 16 else []))
def test_accurate_tier_full_mantissa_weights_over_seeds(wseed):
    """OMGSR-S 128->512 at SD2.1 shapes with weights that carry FULL fp32 mantissas (nothing pre-rounded to a 16-bit-representable
    value: what a checkpoint looks like after the reference's fp32 LoRA merge, infer/omgsr_s_infer_model.py:16-23), three weight
    draws x two input / noise draws, the SHIPPED precision policy (omgsr_amd/precision.py: operand AND weight two-term splits):
    every case must sit under 8e-4 against the fp32 CPU oracle holding the same weights."""
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef
    torch.set_num_threads(min(16, torch.get_num_threads()))
    vae = seeded_init_(R.AutoencoderKL(), 1101 + 17 * wseed, rounded=False).eval()
    unet = seeded_init_(R.UNet2DConditionModel(), 2202 + 17 * wseed, rounded=False).eval()
    alpha = R.DDPMScheduler().alphas_cumprod[273]
    try:
        pv, pu = AutoencoderKL(), UNet2DConditionModel()
        pv.load_state_dict(vae.state_dict()); pu.load_state_dict(unet.state_dict())
        pipe = OMGSR_S_Infer(None, None, 273, DEV, torch.float32, vae=pv, unet=pu)
        # OMGSR_TEST_TIER=fallback: the same sweep in the forced range-fallback tier (bf16 operands, everything split, split q / k / P / V in the
        # flash kernel; profiles/r06_robustness_range_fallback_*.log records a 12 x 2-draw run)
        tier = os.environ.get("OMGSR_TEST_TIER", "accurate")
        if tier == "fallback":
            pipe.range_fallback.enter()
        worst = 0.0
        for xseed in (0, 1):
            g = torch.Generator().manual_seed(5000 + 10 * wseed + xseed)
            x = synthetic_lq(1, 512, 512, seed=777 + 10 * wseed + xseed)
            prompt = torch.randn(1, 77, 1024, generator=g)               # full-mantissa prompt embeddings too
            eps = torch.randn(1, 4, 64, 64, generator=g)
            vae.posterior_noise = eps
            with torch.no_grad():
                ref = OmgsrSRef(vae, unet, alpha, 273)(x, prompt, 64, 32)
                pipe.vae.posterior_noise = eps.to(DEV)
                got, _ = pipe(x.to(DEV), prompt.to(DEV), 64, 32)
            got = got.float().cpu()
            e, p = rel_l2(got, ref), psnr(got, ref)
            print(f"OMGSR-S 128->512 {tier} tier, full-mantissa weights, weight draw {wseed} input draw {xseed}: rel-L2 {e:.3e} PSNR {p:.1f} dB")
            assert torch.isfinite(got).all() and p >= NORTH_STAR_PSNR
            worst = max(worst, e)
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    assert worst <= ROBUST_REL_L2, worst


# The headline workload itself (BASELINE configs[2]: 256 -> 1024, tiled VAE enc 256 / dec 64, latent tiles 64 / 32) with full-mantissa
# weights; one draw in the suite, OMGSR_ROBUST_DRAWS_1024=<n> for more (DESIGN.md §4 records three)
@pytest.mark.parametrize("draw", list(range(int(os.environ.get("OMGSR_ROBUST_DRAWS_1024", "1")))))
def test_accurate_tier_s1024_tiled_full_mantissa_weights(draw):
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef, TiledVaeRef
    torch.set_num_threads(min(16, torch.get_num_threads()))
    vae = seeded_init_(R.AutoencoderKL(), 3301 + 13 * draw, rounded=False).eval()
    unet = seeded_init_(R.UNet2DConditionModel(), 4402 + 13 * draw, rounded=False).eval()
    alpha = R.DDPMScheduler().alphas_cumprod[273]
    g = torch.Generator().manual_seed(6000 + draw)
    x = synthetic_lq(1, 1024, 1024, seed=888 + draw)
    prompt = torch.randn(1, 77, 1024, generator=g)
    eps = torch.randn(1, 4, 128, 128, generator=g)
    vae.posterior_noise = eps
    try:
        pv, pu = AutoencoderKL(), UNet2DConditionModel()
        pv.load_state_dict(vae.state_dict()); pu.load_state_dict(unet.state_dict())
        pipe = OMGSR_S_Infer(None, None, 273, DEV, torch.float32, vae=pv, unet=pu)
        pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
        pipe.vae.posterior_noise = eps.to(DEV)
        with torch.no_grad():
            ref = OmgsrSRef(TiledVaeRef(vae, 256, 64), unet, alpha, 273)(x, prompt, 64, 32)
            got, _ = pipe(x.to(DEV), prompt.to(DEV), 64, 32)
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    got = got.float().cpu()
    e, p = rel_l2(got, ref), psnr(got, ref)
    print(f"OMGSR-S 256->1024 tiled VAE, accurate tier, full-mantissa weights, draw {draw}: rel-L2 {e:.3e} PSNR {p:.1f} dB")
    assert torch.isfinite(got).all() and p >= NORTH_STAR_PSNR and e <= ROBUST_REL_L2


def test_omgsr_s_512_batch8_bf16_vs_oracle(s_oracle):
    """BASELINE configs[1] AS STATED: OMGSR-S 128->512, batch 8, bf16 (the dispatcher picks kernels by row count, so batch 8 takes
    other paths than batch 1): images 0 and 7 of the batch against the fp32 CPU oracle run on each alone."""
    from omgsr_amd import ops
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef
    vae, unet = seeded_init_(R.AutoencoderKL(), 101).eval(), seeded_init_(R.UNet2DConditionModel(), 202).eval()
    x = synthetic_lq(8, 512, 512, seed=4242)
    eps = torch.randn(8, 4, 64, 64, generator=torch.Generator().manual_seed(7))
    alpha = R.DDPMScheduler().alphas_cumprod[273]
    refs = {}
    with torch.no_grad():
        for i in (0, 7):
            vae.posterior_noise = eps[i:i + 1]
            refs[i] = OmgsrSRef(vae, unet, alpha, 273)(x[i:i + 1], s_oracle["prompt"], 64, 32)
    try:
        pipe = _pipe(s_oracle, torch.bfloat16)
        pipe.vae.posterior_noise = eps.to(DEV)
        with torch.no_grad():
            got, _ = pipe(x.to(device=DEV, dtype=torch.bfloat16), s_oracle["prompt"].to(device=DEV, dtype=torch.bfloat16), 64, 32)
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    assert tuple(got.shape) == (8, 3, 512, 512) and got.dtype == torch.bfloat16
    for i in (0, 7):
        g = got[i:i + 1].float().cpu()
        e, p = rel_l2(g, refs[i]), psnr(g, refs[i])
        print(f"OMGSR-S 128->512 batch 8 bf16, image {i}: rel-L2 {e:.3e} PSNR {p:.1f} dB")
        assert e <= 2.5e-2 and p >= 43.0


# ---- trained-like activation statistics --------------------------------------------------------------------------------
def _plant_token_outliers(unet, scale: float, per_block: int, seed: int, sharpen: float = 1.0) -> int:
    """Outlier channels where a trained SD2.1 checkpoint has them and seeded weights do not - the un-normalised tensors the UNet's mixed-precision
    linears read (the LoRA-targeted layers, train/train_omgsr_s.py:89-100): `per_block` GEGLU hidden channels and `per_block` attention value
    channels (self and cross) of every transformer block are scaled by 2^k with the inverse folded into the consuming projection (ff.net.2 /
    to_out.0): the fp32 function is unchanged up to rounding, the hidden / attention-output tensors carry |a| in the hundreds to thousands.
    sharpen > 1 also scales to_q: peaky softmax rows (a trained model's), so an attention output is ~ one value row, not the mean of 4096."""
    g = torch.Generator().manual_seed(seed)
    n = 0
    with torch.no_grad():
        for _, blk in unet.named_modules():
            if not (hasattr(blk, "attn1") and hasattr(blk, "attn2") and hasattr(blk, "ff")):
                continue
            proj, out = blk.ff.net[0].proj, blk.ff.net[2]
            idx = torch.randperm(out.in_features, generator=g)[:per_block]          # the value half of GEGLU: rows [0, inner)
            proj.weight[idx] *= scale; proj.bias[idx] *= scale; out.weight[:, idx] /= scale
            for attn in (blk.attn1, blk.attn2):
                idx = torch.randperm(attn.to_v.out_features, generator=g)[:per_block]
                attn.to_v.weight[idx] *= scale; attn.to_out[0].weight[:, idx] /= scale
                if sharpen != 1.0:
                    attn.to_q.weight *= sharpen
            n += 1
    return n


@pytest.mark.parametrize("scale,sharpen,fallback", [(1024.0, 1.0, "1"), (1024.0, 1.0, "0"), (256.0, 2.0, "1")],
                         ids=["geglu-v-x1024", "geglu-v-x1024-no-fallback", "x256-sharper-softmax"])
def test_accurate_tier_with_planted_outlier_channels(scale, sharpen, fallback, monkeypatch):
    """OMGSR-S 128->512 at SD2.1 shapes, full-mantissa weights, with outlier channels planted in every transformer block (above): GEGLU hidden
    states / attention outputs reach |a| ~ 500 ... 5000, beyond the +-448 of the fixed-scale fp8 correction fields (csrc/common.hip.h). The
    saturation bit must fire; with the fallback (default) the pipeline recomputes with fp16 correction segments on those layers and stays there,
    and the result holds the north-star tolerance against the fp32 oracle with the SAME weights. The no-fallback case records what round 5
    shipped (clamped corrections = single-rounding accuracy on the outlier elements only): it must still hold 1e-3."""
    import warnings
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef
    monkeypatch.setenv("OMGSR_MX_SAT_FALLBACK", fallback)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    vae = seeded_init_(R.AutoencoderKL(), 5101, rounded=False).eval()
    unet = seeded_init_(R.UNet2DConditionModel(), 6202, rounded=False).eval()
    assert _plant_token_outliers(unet, scale, 8, 77, sharpen) == 16
    alpha = R.DDPMScheduler().alphas_cumprod[273]
    g = torch.Generator().manual_seed(9000)
    x = synthetic_lq(1, 512, 512, seed=999)
    prompt = torch.randn(1, 77, 1024, generator=g)
    eps = torch.randn(1, 4, 64, 64, generator=g)
    vae.posterior_noise = eps
    # how large the planted tensors really are (the oracle's own activations)
    peak = {}
    hooks = [m.register_forward_pre_hook(lambda mod, a, k=k: peak.__setitem__(k, max(peak.get(k, 0.0), float(a[0].abs().max()))))
             for k, m in (("ff.net.2 input", unet.down_blocks[0].attentions[0].transformer_blocks[0].ff.net[2]),
                          ("attn1.to_out input", unet.down_blocks[0].attentions[0].transformer_blocks[0].attn1.to_out[0]))]
    try:
        pv, pu = AutoencoderKL(), UNet2DConditionModel()
        pv.load_state_dict(vae.state_dict()); pu.load_state_dict(unet.state_dict())
        pipe = OMGSR_S_Infer(None, None, 273, DEV, torch.float32, vae=pv, unet=pu)
        pipe.vae.posterior_noise = eps.to(DEV)
        with torch.no_grad():
            ref = OmgsrSRef(vae, unet, alpha, 273)(x, prompt, 64, 32)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                got, _ = pipe(x.to(DEV), prompt.to(DEV), 64, 32)
                again, _ = pipe(x.to(DEV), prompt.to(DEV), 64, 32)
        rf = pipe.range_fallback
    finally:
        for h in hooks:
            h.remove()
        ops.set_compute_dtype(torch.bfloat16)
    got = got.float().cpu()
    e, p = rel_l2(got, ref), psnr(got, ref)
    print(f"OMGSR-S 128->512 accurate tier, planted outliers x{scale:g} sharpen {sharpen:g} fallback {fallback}: rel-L2 {e:.3e} PSNR {p:.1f} dB; "
          f"oracle peaks {peak}; saturated calls {rf.mx_saturation_count}, demoted {rf.mx_demoted}, range fallback {rf.sticky}")
    assert peak["ff.net.2 input"] > 448.0, peak
    assert rf.mx_saturation_count >= 1 and any("beyond +-448" in str(w.message) for w in caught)
    assert rf.mx_demoted == (fallback == "1") and not rf.sticky
    assert torch.isfinite(got).all() and p >= NORTH_STAR_PSNR and e <= NORTH_STAR_REL_L2
    if fallback == "1":
        assert rf.mx_saturation_count == 1 and torch.equal(again.float().cpu(), got)      # the second call ran in the demoted form from the start


@pytest.mark.parametrize("config", ["s512", "s1024t"])
def test_range_fallback_tier_full_size_vs_oracle(s_oracle, config):
    """the tier a checkpoint with out-of-fp16-range activations runs (precision.RangeFallback, forced: fp32 stream, bf16 MFMA
    operands, every operand and weight a two-term split) at SD2.1 shapes. Round 5 measured 1.15e-3 on the 256->1024 bench draw - over the
    north-star's 1e-3 - because q, k and P entered the flash kernel as single bf16 values; they are two-term splits now (omgsr_attn_args.q_lo_off /
    k_lo_off / p_split, ops.attn_split): the tier must hold 8e-4 (infer/infer_omgsr_s.py:134-149 is the reference's dtype handling)."""
    from omgsr_amd import ops
    from omgsr_amd.testing import psnr, rel_l2
    x, eps, ref = s_oracle[config]
    try:
        pipe = _pipe(s_oracle, torch.float32)
        pipe.range_fallback.enter()
        assert ops.act_dtype() == torch.bfloat16 and ops.precise() and (ops.attn_split() or os.environ.get("OMGSR_ATTN_SPLIT") == "0")
        if config == "s1024t":
            pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
        pipe.vae.posterior_noise = eps.to(DEV)
        with torch.no_grad():
            got, _ = pipe(x.to(DEV), s_oracle["prompt"].to(DEV), 64, 32)
        assert pipe.range_fallback.sticky
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    which = (0, 3) if config == "s1024t" else (0,)
    for j, i in enumerate(which):
        gi, ri = got[i:i + 1].float().cpu(), ref[j:j + 1]
        e, p = rel_l2(gi, ri), psnr(gi, ri)
        print(f"OMGSR-S {config} range-fallback tier image {i}: rel-L2 {e:.3e}  PSNR {p:.1f} dB (bound {ROBUST_REL_L2:g})")
        assert torch.isfinite(gi).all() and e <= ROBUST_REL_L2 and p >= NORTH_STAR_PSNR

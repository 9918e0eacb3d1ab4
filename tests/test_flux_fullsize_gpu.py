"""BASELINE configs[3] at FULL size and FULL depth: OMGSR-F 256 -> 1024 (infer/omgsr_f_infer_model.py:174-212,322-336) —
the untiled FLUX VAE at 1024^2 (d = 512, N = 16384 mid attention), the 2x2 pack, ONE FluxTransformer2DModel call at
sigma(t* = 244) with all 19 double + 38 single blocks at FLUX.1-dev width (D 3072, 24 heads x 128, 4096 + 512 tokens), the
Euler step, unpack, un-scale and decode — against the fp32 CPU oracle.

The oracle's DiT streams its weights block by block (oracle/flux_streamed_ref.py: the resident fp32 model is 48 GB) from the
product model's own parameters, which are generated ON the GPU with full fp32 mantissas (omgsr_amd.testing.seeded_init_device_:
nothing is pre-rounded to a 16-bit-representable value), so both sides hold bit-identical weights by construction and the
accurate tier is measured on weights it has to round itself. One oracle run (~90 TFLOP on the host) serves every comparison:
the pipeline's image and, through a trace of the oracle's DiT call, the DiT's velocity alone.

Tiers: fp32 = accurate tier at the north-star bar (rel-L2 <= 1e-3, PSNR >= 60 dB); bf16 = the reference's default dtype: the
module is cast to bf16 (weights rounded to 8-bit mantissas, like `--weight_dtype bf16` does) and compared with the SAME fp32
oracle, so its bound includes the weight rounding of the tier (stated below, measured 2026-10 on MI355X).
"""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

BOUNDS = {          # tier: (pipeline rel-L2, pipeline PSNR dB, DiT-alone rel-L2)
    "fp32": (1e-3, 60.0, 1e-3),
    "bf16": (3e-2, 40.0, 2.5e-2),       # measured 1.8e-2 / 44.8 dB / 1.3e-2 (weights rounded to bf16 by the tier, fp32 oracle)
}


@pytest.fixture(scope="module")
def f_case():
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import FLUX_VAE_CONFIG, FluxTransformer2DModel
    from omgsr_amd.testing import seeded_init_, seeded_init_device_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.flux_streamed_ref import StreamedFlux, module_fetcher
    from oracle.pipeline_ref import OmgsrFRef, prepare_latent_image_ids
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ops.set_compute_dtype(torch.float32)
    with torch.device("meta"):
        pf = FluxTransformer2DModel()
    pf = pf.to_empty(device=DEV)
    draw = int(os.environ.get("OMGSR_FLUX_DRAW", "0"))      # other weight / input / noise draws (DESIGN.md §4 records draws 0-2)
    seeded_init_device_(pf, 404 + 31 * draw)           # full-mantissa fp32 weights, generated on the GPU
    ov = seeded_init_(R.AutoencoderKL(**FLUX_VAE_CONFIG), 303 + 31 * draw, rounded=False).eval()
    g = torch.Generator().manual_seed(4321 + draw)
    x = synthetic_lq(1, 1024, 1024, seed=1234 + draw)
    eps = torch.randn(1, 16, 128, 128, generator=torch.Generator().manual_seed(99 + draw))
    pe, pooled = torch.randn(1, 512, 4096, generator=g), torch.randn(1, 768, generator=g)
    tids, iids = torch.zeros(512, 3), prepare_latent_image_ids(64, 64)
    ov.posterior_noise = eps
    st = StreamedFlux(module_fetcher(pf))
    st.trace = []
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = OmgsrFRef(ov, st, 244, 1.0)(x, pe, pooled, tids, iids, 128, 64)
    print(f"fp32 oracle, OMGSR-F 256->1024 with the full-depth DiT streamed: {time.perf_counter() - t0:.1f} s")
    tok, vel = st.trace[0]
    yield dict(flux=pf, vae_sd=ov.state_dict(), x=x, eps=eps, pe=pe, pooled=pooled, tids=tids, iids=iids, ref=ref, tok=tok, vel=vel)
    ops.set_compute_dtype(torch.bfloat16)


def _pipe(c, tier):
    from omgsr_amd.diffusers_api import AutoencoderKL, FLUX_VAE_CONFIG
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer
    wd = torch.float32 if tier == "fp32" else torch.bfloat16
    pv = AutoencoderKL(**FLUX_VAE_CONFIG)
    pv.load_state_dict(c["vae_sd"])
    pf = c["flux"]
    pf.round_timestep_to_weight_dtype = False      # condition on the exact sigma(t*), like the fp32 oracle does
    return OMGSR_F_Infer(None, None, DEV, wd, 244, 1.0, vae=pv, flux_transformer=pf), wd


# order matters: the bf16 legs cast the shared module in place, so both fp32 cases run first
# order matters (pytest runs a file top to bottom): the bf16 legs at the end of the file cast the shared module in place, so the fp32
# cases and the range-fallback case run first
@pytest.mark.parametrize("tier,batch", [("fp32", 1), ("fp32", 8)])
def test_omgsr_f_1024_full_depth_vs_oracle(f_case, tier, batch):
    """batch 1: the pipeline's image and the DiT's velocity alone against the oracle.
    batch 8 = the per-GPU share of BASELINE configs[4] (OMGSR-F 256->1024, batch 64 over 8 GPUs; VERDICT r3 'untested config'): the
    Flux executor flattens the batch to B x 4608-row GEMMs, so igemm_p8_kernel / split-K / attention grids differ from batch 1.
    Images 0 and 7 of the batch ARE the oracle's image (same input, same posterior noise; images 1-6 are other draws), so the ONE
    streamed oracle run checks both ends of the flattened row range; and each must agree with the batch-1 result of the same tier
    up to summation order."""
    from omgsr_amd import ops
    from omgsr_amd.pipelines.omgsr_f import get_flux_setting_timesteps
    from omgsr_amd.testing import psnr, rel_l2, synthetic_lq
    c = f_case
    tol, min_psnr, tol_dit = BOUNDS[tier]
    try:
        pipe, wd = _pipe(c, tier)
        to = lambda t: t.to(device=DEV, dtype=wd)      # noqa: E731
        if batch == 1:
            pipe.vae.posterior_noise = c["eps"].to(DEV)
            with torch.no_grad():
                got, _ = pipe(to(c["x"]), to(c["pe"]), to(c["pooled"]), to(c["tids"]), to(c["iids"]), 128, 64)
                t_curr = get_flux_setting_timesteps()[-(244 + 1)]
                vel = pipe.flux_transformer(hidden_states=to(c["tok"]), timestep=torch.tensor([t_curr], device=DEV),
                                            guidance=torch.full((1,), 1.0, device=DEV), pooled_projections=to(c["pooled"]),
                                            encoder_hidden_states=to(c["pe"]), txt_ids=to(c["tids"]), img_ids=to(c["iids"]), return_dict=False)[0]
        else:
            others = synthetic_lq(6, 1024, 1024, seed=77)
            x8 = torch.cat([c["x"], others, c["x"]], dim=0)
            e8 = torch.cat([c["eps"], torch.randn(6, 16, 128, 128, generator=torch.Generator().manual_seed(5)), c["eps"]], dim=0)
            pipe.vae.posterior_noise = e8.to(DEV)
            with torch.no_grad():
                got8, _ = pipe(to(x8), to(c["pe"]), to(c["pooled"]), to(c["tids"]), to(c["iids"]), 128, 64)
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    if batch == 1:
        got, vel = got.float().cpu(), vel.float().cpu()
        c.setdefault("got_b1", {})[tier] = got
        e, p, ed = rel_l2(got, c["ref"]), psnr(got, c["ref"]), rel_l2(vel, c["vel"])
        print(f"OMGSR-F 256->1024, 19+38 blocks, {tier}: pipeline rel-L2 {e:.3e} PSNR {p:.1f} dB; DiT alone rel-L2 {ed:.3e} "
              f"(bounds {tol:g} / {min_psnr} dB / {tol_dit:g})")
        assert got.shape == c["ref"].shape and torch.isfinite(got).all() and torch.isfinite(vel).all()
        assert ed <= tol_dit and e <= tol and p >= min_psnr
        return
    got8 = got8.float().cpu()
    assert tuple(got8.shape) == (8, 3, 1024, 1024) and torch.isfinite(got8).all()
    b1 = c.get("got_b1", {}).get(tier)
    for i in (0, 7):
        g = got8[i:i + 1]
        e, p = rel_l2(g, c["ref"]), psnr(g, c["ref"])
        d = rel_l2(g, b1) if b1 is not None else float("nan")
        print(f"OMGSR-F 256->1024, 19+38 blocks, {tier}, batch 8, image {i}: rel-L2 {e:.3e} PSNR {p:.1f} dB vs the oracle; vs the batch-1 result {d:.3e}")
        assert e <= tol and p >= min_psnr
        # Batch 1 and batch 8 run the SAME arithmetic up to fp32 summation order (batch 1's small-M GEMMs take split-K, batch 8's do not).
        # That is enough to make them two different realisations of the tier's rounding noise: a sum that lands on the other side of a
        # 16-bit rounding boundary flips an operand by one ulp - a perturbation the size of the noise itself - and the flips cascade
        # until the downstream roundings of the two runs are decorrelated (tools/flux_batch_trace.py: first differing op 5e-6, 1.3e-4
        # one block later, 1.5e-4 after 2 + 2 blocks in the accurate tier / 3.7e-3 in bf16 = ~28 % of either tier's own error; measured
        # here at full depth 4.4e-4 / 1.6e-2 where each run sits 5.5e-4 / 1.8e-2 from the oracle). The distance between two realisations
        # is bounded by the sum of their distances to the oracle, which is what is asserted.
        if b1 is not None:
            assert d <= 2 * tol
    assert rel_l2(got8[1:2], c["ref"]) > 10 * tol           # (the other images really are other images)


def _scale_v_channels(flux, channels, s):
    """Multiply a few VALUE channels of some blocks by 2^s and the matching input columns of the projection that reads the attention
    output by 2^-s: attention is linear in V, powers of two are exact, so the fp32 function (and the oracle's image) is unchanged bit for
    bit while V^T - a 16-bit MFMA operand - now carries outlier channels far beyond fp16's 65504 (what real FLUX activations do)."""
    f = 2.0 ** s
    with torch.no_grad():
        for b in (flux.transformer_blocks[0], flux.transformer_blocks[9]):
            for ch in channels:
                b.attn.to_v.weight[ch].mul_(f); b.attn.to_v.bias[ch].mul_(f)
                b.attn.add_v_proj.weight[ch].mul_(f); b.attn.add_v_proj.bias[ch].mul_(f)
                b.attn.to_out[0].weight[:, ch].mul_(1.0 / f); b.attn.to_add_out.weight[:, ch].mul_(1.0 / f)
        for b in (flux.single_transformer_blocks[3], flux.single_transformer_blocks[20]):
            for ch in channels:
                b.attn.to_v.weight[ch].mul_(f); b.attn.to_v.bias[ch].mul_(f)
                b.proj_out.weight[:, ch].mul_(1.0 / f)          # proj_out reads [attn (D) | mlp]: attention channel ch is input column ch


def test_omgsr_f_1024_full_depth_outlier_channels_take_the_range_fallback(f_case):
    """the guard -> fallback -> result path at FULL depth (19 + 38 blocks at FLUX.1-dev width), not only at the toy scale
    of test_precise_gpu. Three value channels of four blocks are scaled by 2^18 (function unchanged, see _scale_v_channels): the fp16
    operand V^T clips, the range guard fires at forward()'s own synchronisation, the call is recomputed with bf16 operands and every
    operand / weight split, the pipeline stays range-safe - and that result meets the north-star bar against the SAME oracle image."""
    import warnings
    from omgsr_amd import ops
    from omgsr_amd.testing import psnr, rel_l2
    c = f_case
    chans, s = (5, 1033, 2777), 18
    _scale_v_channels(c["flux"], chans, s)
    try:
        pipe, wd = _pipe(c, "fp32")
        to = lambda t: t.to(device=DEV, dtype=wd)      # noqa: E731
        pipe.vae.posterior_noise = c["eps"].to(DEV)
        with torch.no_grad(), warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            got, _ = pipe(to(c["x"]), to(c["pe"]), to(c["pooled"]), to(c["tids"]), to(c["iids"]), 128, 64)
        assert any("65504" in str(w.message) for w in wl)
        assert pipe.range_fallback.count == 1 and pipe.range_fallback.sticky and ops.act_dtype() == torch.bfloat16 and ops.precise()
        with torch.no_grad(), warnings.catch_warnings(record=True) as wl2:
            warnings.simplefilter("always")
            again, _ = pipe(to(c["x"]), to(c["pe"]), to(c["pooled"]), to(c["tids"]), to(c["iids"]), 128, 64)
        assert not any("65504" in str(w.message) for w in wl2) and pipe.range_fallback.count == 1 and torch.equal(again, got)      # ONE range-safe pass
        got = got.float().cpu()
        e, p = rel_l2(got, c["ref"]), psnr(got, c["ref"])
        print(f"OMGSR-F 256->1024, 19+38 blocks, accurate tier with 2^{s} outlier value channels -> range fallback: rel-L2 {e:.3e} PSNR {p:.1f} dB")
        assert torch.isfinite(got).all() and e <= 1e-3 and p >= 60.0
    finally:
        try:
            pipe.range_fallback.reset()
        except NameError:
            pass
        _scale_v_channels(c["flux"], chans, -s)           # exact inverse: the module is shared with the cases below
        ops.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("tier,batch", [("bf16", 1), ("bf16", 8)])
def test_omgsr_f_1024_full_depth_vs_oracle_bf16(f_case, tier, batch):
    test_omgsr_f_1024_full_depth_vs_oracle(f_case, tier, batch)

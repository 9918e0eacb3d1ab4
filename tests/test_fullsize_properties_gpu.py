"""Size-independent properties at BASELINE.json's FULL sizes (SD2.1-base shapes, 512 / 1024 px), where the fp32 CPU oracle
takes minutes (bench.py's cpu_baseline leg does that comparison once per run: PSNR 46 dB bf16 / 64 dB fp16):
batch invariance, scaling linearity of the conv kernels, identity of the Gaussian latent stitch, tiled == untiled VAE
when one tile covers the image, and agreement of the two 16-bit modes with each other."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def full_s():
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import seeded_init_
    ops.set_compute_dtype(torch.bfloat16)
    vae, unet = seeded_init_(AutoencoderKL(), 101), seeded_init_(UNet2DConditionModel(), 202)
    pipe = OMGSR_S_Infer(None, None, 273, DEV, torch.bfloat16, vae=vae, unet=unet)
    g = torch.Generator().manual_seed(7)
    prompt = torch.randn(1, 77, 1024, generator=g).to(torch.bfloat16).to(DEV)
    return pipe, prompt


def _lq(B, side, seed):
    from omgsr_amd import ops
    from omgsr_amd.testing import synthetic_lq
    return ops.nchw_to_nhwc(synthetic_lq(B, side, side, seed=seed).to(DEV), 8)


def test_determinism_and_batch_invariance_full_size(full_s):
    """OMGSR-S 128->512 at SD2.1 shapes: the same batch twice is BIT-identical (fixed-order reductions, no atomics); a
    batch of 3 equals three batch-1 runs up to summation order (the dispatcher picks tile shapes / split-K by problem
    size, so fp32 partial sums associate differently and bf16 roundings flip; through ~100 layers the two runs are two
    independent draws of the bf16 rounding noise around the fp32 result: measured 2.5e-2 = sqrt(2) x the 1.8e-2 of either
    run against the fp32 oracle). Cross-image leakage would be O(1)."""
    from omgsr_amd.testing import rel_l2
    pipe, prompt = full_s
    x = _lq(3, 512, 11)
    eps = torch.randn(3, 4, 64, 64, generator=torch.Generator().manual_seed(12)).to(DEV)
    with torch.no_grad():
        pipe.vae.posterior_noise = eps
        full = pipe.sr_nhwc(x, prompt, 64, 32)
        again = pipe.sr_nhwc(x, prompt, 64, 32)
        singles = []
        for i in range(3):
            pipe.vae.posterior_noise = eps[i:i + 1]
            singles.append(pipe.sr_nhwc(x[i:i + 1].contiguous(), prompt, 64, 32))
    assert torch.isfinite(full.float()).all()
    assert torch.equal(full, again)
    for i in range(3):
        e = rel_l2(full[i:i + 1, ..., :3].float().cpu(), singles[i][..., :3].float().cpu())
        print(f"image {i}: batch-3 vs batch-1 rel-L2 {e:.2e}")
        assert e < 5e-2


def test_latent_stitch_identity_1024(full_s):
    """128 x 128 latent, tile 64 / overlap 32 (the 1024-px configuration): stitching the tiles of an identity denoiser
    returns the latent (Gaussian weights normalise to one) within one 16-bit rounding."""
    from omgsr_amd.pipelines.latent_tiling import tiled_denoise
    z = torch.zeros(2, 128, 128, 8, device=DEV, dtype=torch.bfloat16)
    z[..., :4] = torch.randn(2, 128, 128, 4, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).to(DEV)
    out = tiled_denoise(z, 4, 64, 32, lambda t: t)
    assert (out[..., :4].float() - z[..., :4].float()).abs().max().item() <= 2 ** -7 * z.float().abs().max().item()
    assert bool((out[..., 4:] == 0).all())


@pytest.mark.parametrize("N,C,Cout,H,W", [(4, 128, 128, 512, 512), (4, 512, 512, 128, 128), (36, 320, 320, 64, 64), (36, 1280, 1280, 16, 16)])
def test_conv_scaling_linearity_full_size(N, C, Cout, H, W):
    """conv(2x) == 2 conv(x) bit for bit without a bias (power-of-two scaling commutes with every rounding) on the
    layer shapes of the default bench: halo-tile, LDS-DMA (narrow map) and split paths."""
    from omgsr_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(N, H, W, C, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w = torch.randn(Cout, C, 3, 3, generator=g) * (9 * C) ** -0.5
    pw = ops.pack_conv_weight(w, None, device=DEV)
    y1 = ops.conv2d(x, pw, pad=1)
    y2 = ops.conv2d(x * 2, pw, pad=1)
    assert torch.isfinite(y1.float()).all() and y1.float().abs().max() > 0
    assert torch.equal(y2.float(), y1.float() * 2)


def test_tiled_vae_with_one_tile_equals_untiled(full_s):
    """VAEHook with a tile that covers the image takes the reference's "tiny, unnecessary to tile" branch: bit-identical."""
    pipe, _ = full_s
    from omgsr_amd.pipelines.vaehook import VAEHook
    z = torch.zeros(1, 64, 64, 8, device=DEV, dtype=torch.bfloat16)
    z[..., :4] = torch.randn(1, 64, 64, 4, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        plain = pipe.vae.decoder.run_nhwc(z) if hasattr(pipe.vae.decoder, "run_nhwc") else pipe.vae.decoder.nhwc(z)
        hook = VAEHook(pipe.vae.decoder, 64, is_decoder=True, fast_decoder=False, fast_encoder=False, color_fix=False, to_gpu=True)
        tiled = hook(z)
    assert torch.equal(plain, tiled)


def test_bf16_and_fp16_modes_agree_full_size(full_s):
    """The two 16-bit modes run the same kernels (templates on the element type): on OMGSR-S 128->512 at SD2.1 shapes their
    outputs agree to the bf16 mode's own error level (PSNR > 40 dB), i.e. neither mode has a dtype-specific defect."""
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import psnr, seeded_init_
    pipe, prompt = full_s
    x = _lq(1, 512, 21)
    eps = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(22)).to(DEV)
    with torch.no_grad():
        pipe.vae.posterior_noise = eps
        a = pipe.sr_nhwc(x, prompt, 64, 32).float().cpu()
    try:
        vae, unet = seeded_init_(AutoencoderKL(), 101), seeded_init_(UNet2DConditionModel(), 202)
        p16 = OMGSR_S_Infer(None, None, 273, DEV, torch.float16, vae=vae, unet=unet)          # switches the library to fp16
        p16.vae.posterior_noise = eps
        with torch.no_grad():
            b = p16.sr_nhwc(x.to(torch.float16), prompt.to(torch.float16), 64, 32).float().cpu()
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    assert torch.isfinite(b).all()
    p = psnr(b[..., :3].clamp(-1, 1), a[..., :3].clamp(-1, 1))
    print(f"bf16 vs fp16 mode, OMGSR-S 512: PSNR {p:.1f} dB")
    assert p > 40.0


@pytest.mark.parametrize("wd", [torch.bfloat16, torch.float32], ids=["bf16", "accurate"])
def test_batch_invariant_mode_is_exact_full_size(wd):
    """a batch-B result must equal B independent batch-1 results. With ops.set_batch_invariant(True) the dispatcher's
    kernel-family / split-K choices depend on ONE sample's size, so at the full SD2.1 shapes (OMGSR-S 128->512, latent-tiled UNet
    path included via a 96x96 latent) batch 3 == three batch-1 runs BIT FOR BIT, in the fast and in the accurate tier."""
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import seeded_init_, synthetic_lq
    try:
        ops.set_batch_invariant(True)
        pipe = OMGSR_S_Infer(None, None, 273, DEV, wd, vae=seeded_init_(AutoencoderKL(), 101), unet=seeded_init_(UNet2DConditionModel(), 202))
        prompt = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(7)).to(DEV, wd)
        for side in (512, 768):
            x = synthetic_lq(3, side, side, seed=11).to(DEV, wd)
            eps = torch.randn(3, 4, side // 8, side // 8, generator=torch.Generator().manual_seed(12)).to(DEV)
            with torch.no_grad():
                pipe.vae.posterior_noise = eps
                full, _ = pipe(x, prompt, 64, 32)
                for i in range(3):
                    pipe.vae.posterior_noise = eps[i:i + 1]
                    one, _ = pipe(x[i:i + 1].contiguous(), prompt, 64, 32)
                    assert torch.equal(one, full[i:i + 1]), f"{wd} {side}px image {i}: batch-3 and batch-1 bits differ"
    finally:
        ops.set_batch_invariant(False)
        ops.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("wd", [torch.bfloat16, torch.float32], ids=["bf16", "accurate"])
def test_flux_batch_invariant_mode_is_exact_full_width(wd):
    """The same property for the Flux executor ('no Flux batch-invariance test'), FLUX.1-dev width, 2 + 2 blocks, 4096 + 512
    tokens: with ops.set_batch_invariant(True) a batch of 3 equals three batch-1 calls BIT FOR BIT. Without the switch the two differ by
    ~28 % of the tier's own error (batch 1's small-M GEMMs take split-K; profiles/r04_experiments.md): in batch-invariant mode the entry
    points that put the images on grid.z never split K, and the flattened single-stream GEMMs are dispatched on one image's rows."""
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import get_flux_setting_timesteps, prepare_latent_image_ids
    from omgsr_amd.testing import seeded_init_device_
    try:
        ops.set_compute_dtype(wd)
        ops.set_batch_invariant(True)
        with torch.device("meta"):
            f = FluxTransformer2DModel(num_layers=2, num_single_layers=2)
        f = f.to_empty(device=DEV)
        seeded_init_device_(f, 404)
        f = f.to(wd).eval()
        if wd == torch.float32:
            from omgsr_amd.precision import apply_default_policy
            apply_default_policy(flux=f)
        g = torch.Generator().manual_seed(1)
        tok = torch.randn(3, 4096, 64, generator=g).to(DEV, wd)
        pe, pooled = torch.randn(1, 512, 4096, generator=g).to(DEV, wd), torch.randn(1, 768, generator=g).to(DEV, wd)
        tids, iids = torch.zeros(512, 3, device=DEV, dtype=wd), prepare_latent_image_ids(64, 64, DEV, wd)
        t = torch.tensor([get_flux_setting_timesteps()[-(244 + 1)]], device=DEV)

        def fwd(x):
            with torch.no_grad():
                return f(hidden_states=x, timestep=t, guidance=torch.full((x.shape[0],), 1.0, device=DEV), pooled_projections=pooled,
                         encoder_hidden_states=pe, txt_ids=tids, img_ids=iids, return_dict=False)[0]
        full = fwd(tok)
        for i in range(3):
            one = fwd(tok[i:i + 1].contiguous())
            assert torch.equal(one, full[i:i + 1]), f"{wd} image {i}: batch-3 and batch-1 bits differ"
        ops.set_batch_invariant(False)
        d = ((fwd(tok[:1].contiguous()).float() - fwd(tok)[:1].float()).norm() / full[:1].float().norm()).item()
        print(f"Flux 2+2, {wd}: default dispatch, batch 1 vs batch 3: rel-L2 {d:.3e} (summation order -> decorrelated rounding noise)")
    finally:
        ops.set_batch_invariant(False)
        ops.set_compute_dtype(torch.bfloat16)

"""SURVEY §8(f) f1 on the GPU: omgsr_colorfix (uint8 conversion, AdaIN, wavelet) through the C ABI vs the oracle
(oracle/colorfix_ref.py, pinned to the reference's own functions by tests/test_colorfix_golden.py).

Bar: the plain uint8 conversion is bit-exact. The two colour fixes run a short fp32 chain before the byte TRUNCATION;
the kernel's per-image statistics are integer-exact sums where torch's CPU reduction rounds in fp32, so a value that
lands within an ulp of a byte boundary may truncate to the neighbouring byte: asserted <= 1 LSB on <= 0.5 % of bytes
(measured: 0 - 0.05 %)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=["bf16", "fp16"])
def compute_dtype(request):
    from omgsr_amd import ops
    ops.set_compute_dtype(torch.bfloat16 if request.param == "bf16" else torch.float16)
    yield request.param
    ops.set_compute_dtype(torch.bfloat16)


def _case(B, H, W, seed):
    from omgsr_amd import ops
    g = torch.Generator().manual_seed(seed)
    base = torch.nn.functional.interpolate(torch.rand(B, 3, 5, 7, generator=g), size=(H, W), mode="bicubic", align_corners=False)
    sr = ((base + 0.1 * torch.randn(B, 3, H, W, generator=g)) * 2.2 - 1.1).to(ops.act_dtype())       # some values beyond [-1, 1]
    src = (base * torch.tensor([0.8, 1.0, 1.15]).view(1, 3, 1, 1) + 0.04).clamp(0, 1).mul(255).to(torch.uint8)
    sr_nhwc = torch.zeros(B, H, W, 8, dtype=ops.act_dtype())
    sr_nhwc[..., :3] = sr.permute(0, 2, 3, 1)
    return sr, src, sr_nhwc.to(DEV), src.permute(0, 2, 3, 1).contiguous().to(DEV)


def _cmp(got_hwc, ref_chw, name, exact=True):
    """Byte output: the bar is bit-exact (the kernels keep the reference's fp32 chain op by op, no FMA contraction ahead of the
    byte truncation, integer-exact statistics); `exact=False` is kept only for ad-hoc experiments."""
    got = got_hwc.cpu().permute(0, 3, 1, 2).to(torch.int16)
    ref = ref_chw.to(torch.int16)
    d = (got - ref).abs()
    frac = (d > 0).float().mean().item()
    print(f"{name}: {frac * 100:.4f} % of bytes differ, max {int(d.max())} LSB")
    if exact:
        assert int(d.max()) == 0, name
    else:
        assert int(d.max()) <= 1 and frac <= 5e-3, f"{name}: max {int(d.max())} LSB on {frac * 100:.3f} %"


@pytest.mark.parametrize("B,H,W", [(1, 64, 96), (2, 73, 131), (1, 256, 256)])
def test_uint8_conversion_is_bit_exact(B, H, W):
    from omgsr_amd.colorfix import color_fix
    from oracle import colorfix_ref as R
    sr, src, sr_d, src_d = _case(B, H, W, 5)
    _cmp(color_fix(sr_d, None, "nofix"), R.model_output_to_u8(sr), "uint8 conversion", exact=True)


@pytest.mark.parametrize("B,H,W", [(1, 64, 96), (2, 73, 131), (1, 256, 256)])
def test_adain_color_fix(B, H, W):
    from omgsr_amd.colorfix import adain_color_fix
    from oracle import colorfix_ref as R
    sr, src, sr_d, src_d = _case(B, H, W, 6)
    tgt = R.model_output_to_u8(sr)
    ref = torch.cat([R.adain_color_fix_u8(tgt[i:i + 1], src[i:i + 1]) for i in range(B)])     # the reference fixes image by image
    _cmp(adain_color_fix(sr_d, src_d), ref, "adain")


@pytest.mark.parametrize("B,H,W", [(1, 64, 96), (2, 73, 131), (1, 256, 256), (1, 20, 24)])
def test_wavelet_color_fix(B, H, W):
    from omgsr_amd.colorfix import wavelet_color_fix
    from oracle import colorfix_ref as R
    sr, src, sr_d, src_d = _case(B, H, W, 7)
    tgt = R.model_output_to_u8(sr)
    ref = torch.cat([R.wavelet_color_fix_u8(tgt[i:i + 1], src[i:i + 1]) for i in range(B)])
    _cmp(wavelet_color_fix(sr_d, src_d), ref, "wavelet")


def test_golden_images_through_the_kernel():
    """The reference-captured images (tests/golden/colorfix.npz): sr chosen so that its uint8 conversion IS the golden target."""
    import os
    import numpy as np
    from omgsr_amd import ops
    from omgsr_amd.colorfix import color_fix
    from oracle import colorfix_ref as R
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "colorfix.npz"))
    tgt, src = torch.from_numpy(G["target_u8"]), torch.from_numpy(G["source_u8"])
    # a 16-bit value whose (x*0.5+0.5 -> clip -> *255 -> trunc) reproduces each target byte: search the mid-point
    sr = (((tgt.float() + 0.5) / 255.0) * 2.0 - 1.0).to(ops.act_dtype())
    ok = R.model_output_to_u8(sr) == tgt
    if not bool(ok.all()):
        pytest.skip("golden target bytes are not all representable through this 16-bit dtype")     # bf16: 1/256 steps near 1.0
    nhwc = torch.zeros(1, tgt.shape[2], tgt.shape[3], 8, dtype=ops.act_dtype())
    nhwc[..., :3] = sr.permute(0, 2, 3, 1)
    src_d = src.permute(0, 2, 3, 1).contiguous().to(DEV)
    adain_ref = R.to_pil_u8(torch.from_numpy(G["adain"]).clamp(0, 1))
    wav_ref = R.to_pil_u8(torch.from_numpy(G["wavelet"]).clamp(0, 1))
    _cmp(color_fix(nhwc.to(DEV), src_d, "adain"), adain_ref, "golden adain")
    _cmp(color_fix(nhwc.to(DEV), src_d, "wavelet"), wav_ref, "golden wavelet")


def test_image_to_model_input_is_bit_exact():
    from omgsr_amd import ops
    from omgsr_amd.colorfix import image_to_model_input
    g = torch.Generator().manual_seed(9)
    img = torch.randint(0, 256, (2, 37, 53, 3), generator=g, dtype=torch.uint8)
    ref = (img.float() / 255).to(ops.act_dtype()) * 2 - 1          # F.to_tensor(...).to(dtype) * 2 - 1, op by op in dtype
    got = image_to_model_input(img.to(DEV)).cpu()
    assert torch.equal(got[..., :3], ref) and bool((got[..., 3:] == 0).all())


def test_driver_loop_body_on_device():
    """uint8 in -> model -> uint8 colour-fixed out, against the oracle pipeline + oracle post-process."""
    from omgsr_amd import ops
    from omgsr_amd.colorfix import super_resolve_u8
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import seeded_init_
    from oracle import colorfix_ref as C, diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef
    vcfg = dict(block_out_channels=[32, 64, 128, 128], layers_per_block=1, norm_num_groups=32)
    ucfg = dict(block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4], cross_attention_dim=128, layers_per_block=2)
    ov, ou = seeded_init_(R.AutoencoderKL(**vcfg), 41).eval(), seeded_init_(R.UNet2DConditionModel(**ucfg), 42).eval()
    pv, pu = AutoencoderKL(**vcfg), UNet2DConditionModel(**ucfg)
    pv.load_state_dict(ov.state_dict()); pu.load_state_dict(ou.state_dict())
    g = torch.Generator().manual_seed(43)
    img = torch.nn.functional.interpolate(torch.rand(1, 3, 8, 8, generator=g), size=(128, 128), mode="bicubic").clamp(0, 1).mul(255).to(torch.uint8)
    ehs = torch.randn(1, 77, 128, generator=g).to(torch.bfloat16).float()
    eps = torch.randn(1, 4, 16, 16, generator=g)
    ov.posterior_noise = eps; pv.posterior_noise = eps
    pipe = OMGSR_S_Infer(None, None, 273, DEV, ops.act_dtype(), vae=pv, unet=pu)
    got = super_resolve_u8(pipe, img.permute(0, 2, 3, 1).contiguous().to(DEV), ehs.to(DEV), 16, 8, align_method="adain")
    lq = (img.float() / 255) * 2 - 1
    with torch.no_grad():
        ref_img = OmgsrSRef(ov, ou, R.DDPMScheduler().alphas_cumprod[273], 273)(lq, ehs, 16, 8)
    ref = C.adain_color_fix_u8(C.model_output_to_u8(ref_img), img)
    d = (got.cpu().permute(0, 3, 1, 2).to(torch.int16) - ref.to(torch.int16)).abs().float()
    print(f"driver loop: mean |diff| {d.mean():.3f} LSB, max {int(d.max())}")
    assert d.mean() < 4.0          # 16-bit model vs fp32 oracle: a few grey levels on a random-weight net

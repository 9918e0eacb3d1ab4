"""SURVEY §8(f) f3 on the GPU: constants exported by one pipeline, loaded into a FRESH one (same weights), give bit-identical
outputs without that pipeline ever folding a time embedding or projecting the prompt — both families, both tiers."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
SMALL_VAE = dict(block_out_channels=[32, 64, 128, 128], layers_per_block=1, norm_num_groups=32)
SMALL_UNET = dict(block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4], cross_attention_dim=128, layers_per_block=2)
SMALL_FLUX = dict(num_layers=2, num_single_layers=3, num_attention_heads=2, attention_head_dim=128, joint_attention_dim=64,
                  pooled_projection_dim=32, in_channels=64)
SMALL_FLUX_VAE = dict(SMALL_VAE, latent_channels=16, use_quant_conv=False, use_post_quant_conv=False, scaling_factor=0.3611, shift_factor=0.1159)


@pytest.fixture(params=[torch.bfloat16, torch.float32], ids=["bf16", "accurate"])
def wd(request):
    from omgsr_amd import ops
    yield request.param
    ops.set_compute_dtype(torch.bfloat16)


def _forbid_folding(monkeypatch, *objs):
    for o, name in objs:
        monkeypatch.setattr(o, name, lambda *a, **k: pytest.fail(f"{name} ran: the constants were not served from the file"))


def test_omgsr_s_constants_round_trip(tmp_path, wd, monkeypatch):
    from omgsr_amd import constants as K
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.diffusers_api.unet_2d_condition import TimestepEmbedding
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import seeded_init_, synthetic_lq

    def make():
        return OMGSR_S_Infer(None, None, 273, DEV, wd, vae=seeded_init_(AutoencoderKL(**SMALL_VAE), 1), unet=seeded_init_(UNet2DConditionModel(**SMALL_UNET), 2))
    g = torch.Generator().manual_seed(5)
    x = synthetic_lq(2, 128, 128).to(DEV)
    eps = torch.randn(2, 4, 16, 16, generator=g).to(DEV)
    prompt = torch.randn(1, 77, 128, generator=g).to(DEV, wd)
    a = make()
    a.vae.posterior_noise = eps
    with torch.no_grad():
        ref, _ = a(x, prompt, 16, 8)
    path = str(tmp_path / "s.safetensors")
    K.export_s(a, prompt, path)
    b = make()
    b.vae.posterior_noise = eps
    loaded_prompt = K.load_s(b, path)
    monkeypatch.setattr(TimestepEmbedding, "fp32", lambda *a_, **k: pytest.fail("the time embedding was folded again"))
    for _, at in K._s_cross_attention(b.unet):
        _forbid_folding(monkeypatch, (at.to_k, "packed"), (at.to_v, "packed"))
    with torch.no_grad():
        got, _ = b(x, loaded_prompt, 16, 8)
    assert torch.equal(got, ref)


def test_omgsr_f_constants_round_trip(tmp_path, wd, monkeypatch):
    from omgsr_amd import constants as K
    from omgsr_amd.diffusers_api import AutoencoderKL, FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer, prepare_latent_image_ids
    from omgsr_amd.testing import seeded_init_, synthetic_lq

    def make():
        return OMGSR_F_Infer(None, None, DEV, wd, 244, 1.0, vae=seeded_init_(AutoencoderKL(**SMALL_FLUX_VAE), 31),
                             flux_transformer=seeded_init_(FluxTransformer2DModel(**SMALL_FLUX), 32))
    g = torch.Generator().manual_seed(6)
    x = synthetic_lq(1, 128, 128).to(DEV)
    eps = torch.randn(1, 16, 16, 16, generator=g).to(DEV)
    pe, pooled = torch.randn(1, 24, 64, generator=g).to(DEV, wd), torch.randn(1, 32, generator=g).to(DEV, wd)
    tids, iids = torch.zeros(24, 3, device=DEV, dtype=wd), prepare_latent_image_ids(8, 8, DEV, wd)
    a = make()
    a.vae.posterior_noise = eps
    with torch.no_grad():
        ref, _ = a(x, pe, pooled, tids, iids, 16, 8)
    path = str(tmp_path / "f.safetensors")
    K.export_f(a, pe, pooled, tids, iids, path)
    d = K.describe(path)
    assert d["metadata"]["family"] == "F" and "flux.mod.single.2.g" in d["tensors"] and "flux.rope.cos" in d["tensors"]
    b = make()
    b.vae.posterior_noise = eps
    args = K.load_f(b, path)
    _forbid_folding(monkeypatch, (b.flux_transformer.time_text_embed, "fp32"), (b.flux_transformer.context_embedder, "packed"))
    with torch.no_grad():
        got, _ = b(x, *args, 16, 8)
    assert torch.equal(got, ref)

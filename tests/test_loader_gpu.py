"""SURVEY §8 rows a1 / a10 / f3 / f4 end to end on the GPU, through the reference's own constructor route
(infer/omgsr_s_infer_model.py:9-32, infer/omgsr_f_infer_model.py:97-133): an HF directory on disk (sharded index + the SD2.1 VAE's
legacy attention key names) and PEFT adapter directories with the reference's names (`vae_encoder_lora_adapter` /
`unet_lora_adapter`; `flux_adapter` / `vae_encoder_adapter`) -> `OMGSR_S_Infer(sd_path, lora_path, 273, device, dtype)` /
`OMGSR_F_Infer(flux_path, lora_path, device, dtype)` -> HIP output compared with the fp32 CPU ORACLE holding the same base
weights with the same adapters merged by the oracle's own LoRA merge; then the constant cache (f3) exported by that pipeline is
loaded into a second pipeline built from the same directories and must reproduce the first one's output bit for bit - and,
being the same tensor, the oracle comparison. Reduced configurations (the full-size comparisons live in
tests/test_fullsize_parity_gpu.py / tests/test_flux_fullsize_gpu.py); base weights and adapters carry full fp32 mantissas."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
SMALL_VAE = dict(block_out_channels=[32, 64, 128, 128], layers_per_block=1, norm_num_groups=32)
SMALL_UNET = dict(block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4], cross_attention_dim=128, layers_per_block=2)
SMALL_FLUX = dict(num_layers=2, num_single_layers=3, num_attention_heads=2, attention_head_dim=128, joint_attention_dim=64,
                  pooled_projection_dim=32, in_channels=64)
SMALL_FLUX_VAE = dict(SMALL_VAE, latent_channels=16, use_quant_conv=False, use_post_quant_conv=False, scaling_factor=0.3611, shift_factor=0.1159)
R_LORA = 4


@pytest.fixture(params=[torch.float32, torch.bfloat16], ids=["accurate", "bf16"])
def wd(request):
    from omgsr_amd import ops
    yield request.param
    ops.set_compute_dtype(torch.bfloat16)


def _adapter(module, targets, seed):
    """A LoRA adapter state dict in PEFT's layout for the Conv2d / Linear modules of `module` whose name ends with a target."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, m in module.named_modules():
        if not any(name.endswith(t) for t in targets) or not hasattr(m, "weight") or m.weight.dim() < 2:
            continue
        w = m.weight
        if w.dim() == 4:
            sd[f"base_model.model.{name}.lora_A.weight"] = torch.randn(R_LORA, w.shape[1], w.shape[2], w.shape[3], generator=g) * 0.05
            sd[f"base_model.model.{name}.lora_B.weight"] = torch.randn(w.shape[0], R_LORA, 1, 1, generator=g) * 0.05
        else:
            sd[f"base_model.model.{name}.lora_A.weight"] = torch.randn(R_LORA, w.shape[1], generator=g) * 0.05
            sd[f"base_model.model.{name}.lora_B.weight"] = torch.randn(w.shape[0], R_LORA, generator=g) * 0.05
    assert sd
    return sd


def _write_adapter(path, sd, targets):
    from safetensors.torch import save_file
    os.makedirs(path)
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump(dict(r=R_LORA, lora_alpha=R_LORA, target_modules=list(targets), peft_type="LORA"), f)
    save_file(sd, os.path.join(path, "adapter_model.safetensors"))


def _legacy_vae_keys(root):
    """Rewrite <root>/vae's checkpoint with the SD2.1-base VAE's legacy attention key names (query / key / value / proj_attn)."""
    from safetensors.torch import load_file, save_file
    f = os.path.join(root, "vae", "diffusion_pytorch_model.safetensors")
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    out = {}
    for k, t in load_file(f).items():
        for new, old in ren.items():
            k = k.replace(f".attentions.0.{new}.", f".attentions.0.{old}.")
        out[k] = t.contiguous()
    assert any(".query." in k for k in out)
    save_file(out, f)


def _tol(wd):
    return (1e-3, 60.0) if wd == torch.float32 else (3e-2, 40.0)


def test_omgsr_s_from_directories_vs_oracle(tmp_path, wd):
    from omgsr_amd import constants as K, ops
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrSRef
    sd_path, lora_path = str(tmp_path / "sd21"), str(tmp_path / "lora")
    ov = seeded_init_(R.AutoencoderKL(**SMALL_VAE), 11, rounded=False).eval()
    ou = seeded_init_(R.UNet2DConditionModel(**SMALL_UNET), 12, rounded=False).eval()
    v, u = AutoencoderKL(**SMALL_VAE), UNet2DConditionModel(**SMALL_UNET)
    v.load_state_dict(ov.state_dict()); u.load_state_dict(ou.state_dict())
    v.save_pretrained(sd_path, subfolder="vae")
    _legacy_vae_keys(sd_path)
    u.save_pretrained(sd_path, subfolder="unet", max_shard_size=8 << 20)                     # sharded + index.json
    assert os.path.isfile(os.path.join(sd_path, "unet", "diffusion_pytorch_model.safetensors.index.json"))
    os.makedirs(os.path.join(sd_path, "scheduler"))
    with open(os.path.join(sd_path, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(_class_name="DDPMScheduler", beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                       num_train_timesteps=1000, prediction_type="epsilon"), f)
    # adapters on the reference's target sets (train/train_omgsr_s.py:61-72,89-100): encoder convs; UNet convs + attention / FF linears
    enc_t = ("conv1", "conv2", "conv_in", "conv_shortcut", "conv", "conv_out", "to_k", "to_q", "to_v", "to_out.0")
    unet_t = ("to_k", "to_q", "to_v", "to_out.0", "conv", "conv1", "conv2", "conv_shortcut", "conv_out", "proj_in", "proj_out", "ff.net.2", "ff.net.0.proj")
    enc_sd, unet_sd = _adapter(ov.encoder, enc_t, 21), _adapter(ou, unet_t, 22)
    _write_adapter(os.path.join(lora_path, "vae_encoder_lora_adapter"), enc_sd, enc_t)
    _write_adapter(os.path.join(lora_path, "unet_lora_adapter"), unet_sd, unet_t)
    assert R.merge_lora_(ov.encoder, enc_sd, R_LORA, R_LORA) == len(enc_sd) // 2 and R.merge_lora_(ou, unet_sd, R_LORA, R_LORA) == len(unet_sd) // 2

    g = torch.Generator().manual_seed(31)
    x = synthetic_lq(2, 192, 192)
    prompt = torch.randn(1, 77, 128, generator=g)
    eps = torch.randn(2, 4, 24, 24, generator=g)
    ov.posterior_noise = eps
    with torch.no_grad():
        ref = OmgsrSRef(ov, ou, R.DDPMScheduler().alphas_cumprod[273], 273)(x, prompt, 16, 8)      # latent 24 > tile 16: tiled UNet

    def make():
        p = OMGSR_S_Infer(sd_path, lora_path, 273, DEV, wd)
        p.vae.posterior_noise = eps.to(DEV)
        return p
    a = make()
    assert float(a.alpha_t) == 0.6357423067092896
    pr = prompt.to(DEV, wd)
    with torch.no_grad():
        got, _ = a(x.to(DEV, wd), pr, 16, 8)
    tol, min_psnr = _tol(wd)
    e, p = rel_l2(got, ref), psnr(got, ref)
    print(f"OMGSR_S_Infer(sd_path, lora_path) {wd}: rel-L2 {e:.3e} PSNR {p:.1f} dB vs the oracle with the same adapters merged")
    assert e <= tol and p >= min_psnr
    cpath = str(tmp_path / "s_constants.safetensors")
    K.export_s(a, pr, cpath)
    b = make()
    with torch.no_grad():
        got_b, _ = b(x.to(DEV, wd), K.load_s(b, cpath), 16, 8)
    assert torch.equal(got_b, got)                                     # served from the file == folded in process ...
    assert rel_l2(got_b, ref) <= tol                                   # ... and therefore the same distance from the ORACLE
    if wd == torch.float32:                                            # a cache folded under another precision policy is refused
        from omgsr_amd import precision
        precision.set_weight_split(b.unet, [])
        with pytest.raises(K.ConstantsMismatch, match="policy"):
            K.load_s(b, cpath)


def test_omgsr_f_from_directories_vs_oracle(tmp_path, wd):
    from omgsr_amd import constants as K
    from omgsr_amd.diffusers_api import AutoencoderKL, FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer
    from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq
    from oracle import diffusers_ref as R
    from oracle.pipeline_ref import OmgsrFRef, prepare_latent_image_ids
    flux_path, lora_path = str(tmp_path / "flux"), str(tmp_path / "lora")
    ov = seeded_init_(R.AutoencoderKL(**SMALL_FLUX_VAE), 41, rounded=False).eval()
    of = seeded_init_(R.FluxTransformer2DModel(**SMALL_FLUX), 42, rounded=False).eval()
    v, fl = AutoencoderKL(**SMALL_FLUX_VAE), FluxTransformer2DModel(**SMALL_FLUX)
    v.load_state_dict(ov.state_dict()); fl.load_state_dict(of.state_dict())
    v.save_pretrained(flux_path, subfolder="vae")
    fl.save_pretrained(flux_path, subfolder="transformer", max_shard_size=6 << 20)           # FLUX.1-dev ships 3 shards + index
    assert os.path.isfile(os.path.join(flux_path, "transformer", "diffusion_pytorch_model.safetensors.index.json"))
    # adapters on the reference's target sets (train/train_omgsr_f.py:132-143,155-169)
    flux_t = ("to_k", "to_q", "to_v", "to_out.0", "add_k_proj", "add_q_proj", "add_v_proj", "to_add_out", "ff.net.0.proj", "ff.net.2",
              "ff_context.net.0.proj", "ff_context.net.2", "proj_mlp", "proj_out", "x_embedder", "norm1.linear", "norm1_context.linear", "norm.linear")
    enc_t = ("conv1", "conv2", "conv_in", "conv_shortcut", "conv", "conv_out", "to_k", "to_q", "to_v", "to_out.0")
    flux_sd, enc_sd = _adapter(of, flux_t, 51), _adapter(ov.encoder, enc_t, 52)
    _write_adapter(os.path.join(lora_path, "flux_adapter"), flux_sd, flux_t)
    _write_adapter(os.path.join(lora_path, "vae_encoder_adapter"), enc_sd, enc_t)
    R.merge_lora_(of, flux_sd, R_LORA, R_LORA); R.merge_lora_(ov.encoder, enc_sd, R_LORA, R_LORA)

    g = torch.Generator().manual_seed(61)
    B, t, Lc = 2, 16, 24
    x = synthetic_lq(B, t * 8, t * 8)
    pe, pooled = torch.randn(1, Lc, 64, generator=g), torch.randn(1, 32, generator=g)
    tids, iids = torch.zeros(Lc, 3), prepare_latent_image_ids(t // 2, t // 2)
    eps = torch.randn(B, 16, t, t, generator=g)
    ov.posterior_noise = eps
    with torch.no_grad():
        ref = OmgsrFRef(ov, of, 244, 1.0)(x, pe, pooled, tids, iids, t, t // 2)

    def make():
        p = OMGSR_F_Infer(flux_path, lora_path, DEV, wd)
        p.flux_transformer.round_timestep_to_weight_dtype = False      # condition on the exact sigma(t*) like the fp32 oracle
        p.vae.posterior_noise = eps.to(DEV)
        return p
    a = make()
    to = lambda z: z.to(DEV, wd)                                       # noqa: E731
    args = (to(pe), to(pooled), to(tids), to(iids))
    with torch.no_grad():
        got, _ = a(to(x), *args, t, t // 2)
    tol, min_psnr = _tol(wd)
    e, p = rel_l2(got, ref), psnr(got, ref)
    print(f"OMGSR_F_Infer(flux_path, lora_path) {wd}: rel-L2 {e:.3e} PSNR {p:.1f} dB vs the oracle with the same adapters merged")
    assert e <= tol and p >= min_psnr
    cpath = str(tmp_path / "f_constants.safetensors")
    K.export_f(a, *args, cpath)
    b = make()
    with torch.no_grad():
        got_b, _ = b(to(x), *K.load_f(b, cpath), t, t // 2)
    assert torch.equal(got_b, got) and rel_l2(got_b, ref) <= tol

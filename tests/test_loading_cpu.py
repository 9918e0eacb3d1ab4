"""Weight plumbing on CPU: HF directory round trip (config.json + safetensors) and the
PeftModel stand-in (adapter dir -> in-place LoRA merge) against the oracle's merge."""
import json
import os

import torch

from oracle import diffusers_ref as R

SMALL_UNET = dict(block_out_channels=[32, 64, 64, 64], attention_head_dim=[1, 2, 2, 2], cross_attention_dim=64)
SMALL_VAE = dict(block_out_channels=[32, 32, 64, 64], layers_per_block=1)


def test_from_pretrained_round_trip(tmp_path):
    from omgsr_amd.diffusers_api import AutoencoderKL, DDPMScheduler, UNet2DConditionModel
    from omgsr_amd.testing import seeded_init_
    root = str(tmp_path / "sd")
    u = seeded_init_(UNet2DConditionModel(**SMALL_UNET), 3)
    v = seeded_init_(AutoencoderKL(**SMALL_VAE), 4)
    u.save_pretrained(root, subfolder="unet")
    v.save_pretrained(root, subfolder="vae")
    os.makedirs(os.path.join(root, "scheduler"))
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(_class_name="DDPMScheduler", beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                       num_train_timesteps=1000, prediction_type="epsilon"), f)
    u2 = UNet2DConditionModel.from_pretrained(root, subfolder="unet")
    v2 = AutoencoderKL.from_pretrained(root, subfolder="vae")
    assert u2.config.block_out_channels == SMALL_UNET["block_out_channels"] and not u2.training
    for a, b in ((u, u2), (v, v2)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    assert DDPMScheduler.from_pretrained(root, subfolder="scheduler").alphas_cumprod[273].item() == 0.6357423067092896
    # the oracle accepts the same state dict (diffusers key names on both sides)
    R.UNet2DConditionModel(**SMALL_UNET).load_state_dict(u.state_dict())
    R.AutoencoderKL(**SMALL_VAE).load_state_dict(v.state_dict())


def _write_adapter(path, sd, r):
    from safetensors.torch import save_file
    os.makedirs(path)
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump(dict(r=r, lora_alpha=r, target_modules=["conv1", "to_q"]), f)
    save_file(sd, os.path.join(path, "adapter_model.safetensors"))


def test_peft_merge_matches_oracle(tmp_path):
    from omgsr_amd.diffusers_api import PeftModel, UNet2DConditionModel
    from omgsr_amd.testing import seeded_init_
    r = 4
    g = torch.Generator().manual_seed(5)
    u = seeded_init_(UNet2DConditionModel(**SMALL_UNET), 3)
    o = R.UNet2DConditionModel(**SMALL_UNET)
    o.load_state_dict(u.state_dict())
    conv = "down_blocks.0.resnets.0.conv1"                       # Conv2d target (train/train_omgsr_s.py:89-100)
    lin = "mid_block.attentions.0.transformer_blocks.0.attn1.to_q"
    cw, lw = dict(u.named_modules())[conv].weight, dict(u.named_modules())[lin].weight
    sd = {f"base_model.model.{conv}.lora_A.weight": torch.randn(r, cw.shape[1], 3, 3, generator=g) * 0.1,
          f"base_model.model.{conv}.lora_B.weight": torch.randn(cw.shape[0], r, 1, 1, generator=g) * 0.1,
          f"base_model.model.{lin}.lora_A.weight": torch.randn(r, lw.shape[1], generator=g) * 0.1,
          f"base_model.model.{lin}.lora_B.weight": torch.randn(lw.shape[0], r, generator=g) * 0.1}
    _write_adapter(str(tmp_path / "unet_lora_adapter"), sd, r)
    before = cw.detach().clone()
    wrapped = PeftModel.from_pretrained(u, str(tmp_path / "unet_lora_adapter"))
    assert wrapped.config.in_channels == 4 and wrapped.dtype == torch.float32      # attribute forwarding (reference :71,76)
    wrapped.merge_and_unload()                                                       # return value discarded upstream
    assert R.merge_lora_(o, sd, r, r) == 2
    assert not torch.equal(cw, before)
    for k, v in u.state_dict().items():
        torch.testing.assert_close(v, o.state_dict()[k], rtol=0, atol=1e-6)
    wrapped.merge_and_unload()                                                       # idempotent
    torch.testing.assert_close(dict(u.named_modules())[conv].weight, dict(o.named_modules())[conv].weight, rtol=0, atol=1e-6)


def test_sharded_checkpoint_with_index_json(tmp_path):
    """FLUX.1-dev's transformer ships as `diffusion_pytorch_model-0000X-of-00003.safetensors` + an index.json weight_map
    (infer/omgsr_f_infer_model.py:99-106 loads it through from_pretrained): shards written by save_pretrained(max_shard_size)
    in diffusers' layout load back bit-exactly, and a shard missing from the map is an error, not a silent partial load."""
    from omgsr_amd.diffusers_api import FluxTransformer2DModel
    from omgsr_amd.testing import seeded_init_
    cfg = dict(num_layers=1, num_single_layers=2, num_attention_heads=2, attention_head_dim=128, joint_attention_dim=64,
               pooled_projection_dim=32, in_channels=64)
    m = seeded_init_(FluxTransformer2DModel(**cfg), 7)
    root = str(tmp_path / "flux")
    m.save_pretrained(root, subfolder="transformer", max_shard_size=6 << 20)
    files = sorted(os.listdir(os.path.join(root, "transformer")))
    shards = [f for f in files if f.endswith(".safetensors")]
    assert "diffusion_pytorch_model.safetensors.index.json" in files and len(shards) >= 3 and "diffusion_pytorch_model.safetensors" not in files
    idx = json.load(open(os.path.join(root, "transformer", "diffusion_pytorch_model.safetensors.index.json")))
    assert set(idx["weight_map"]) == set(m.state_dict()) and set(idx["weight_map"].values()) == set(shards)
    m2 = FluxTransformer2DModel.from_pretrained(root, subfolder="transformer")
    sa, sb = m.state_dict(), m2.state_dict()
    assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    # drop one shard from the map: its keys go missing -> from_pretrained raises
    victim = shards[1]
    idx["weight_map"] = {k: v for k, v in idx["weight_map"].items() if v != victim}
    json.dump(idx, open(os.path.join(root, "transformer", "diffusion_pytorch_model.safetensors.index.json"), "w"))
    import pytest
    with pytest.raises(RuntimeError, match="missing keys"):
        FluxTransformer2DModel.from_pretrained(root, subfolder="transformer")


def test_legacy_vae_attention_keys_are_converted(tmp_path):
    """stabilityai/stable-diffusion-2-1-base/vae stores the mid-block attention as query / key / value / proj_attn (diffusers
    renames them at load time); from_pretrained must load that checkpoint the reference loads (infer/omgsr_s_infer_model.py:11)."""
    from safetensors.torch import save_file
    from omgsr_amd.diffusers_api import AutoencoderKL
    from omgsr_amd.testing import seeded_init_
    v = seeded_init_(AutoencoderKL(**SMALL_VAE), 4)
    root = str(tmp_path / "sd")
    v.save_pretrained(root, subfolder="vae")
    sd = dict(v.state_dict())
    legacy = {}
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, t in sd.items():
        nk = k
        for new, old in ren.items():
            if f".attentions.0.{new}." in k:
                nk = k.replace(f".attentions.0.{new}.", f".attentions.0.{old}.")
                if old == "proj_attn" and k.endswith("weight") and "decoder" in k:
                    t = t[:, :, None, None]            # the 1x1-conv form some checkpoints keep
        legacy[nk] = t.contiguous()
    assert any(".query." in k for k in legacy) and not any(".to_q." in k for k in legacy)
    save_file(legacy, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    v2 = AutoencoderKL.from_pretrained(root, subfolder="vae")
    sb = v2.state_dict()
    assert list(sd) == list(sb) and all(torch.equal(sd[k], sb[k]) for k in sd)


def test_flux_timestep_rounding_follows_the_weight_dtype():
    """diffusers: `timestep.to(hidden_states.dtype) * 1000` (SURVEY C-7). bf16 weights see 504.0, fp16 505.0, fp32 505.11..."""
    import torch
    vals = {}
    for wd in (torch.bfloat16, torch.float16):
        vals[wd] = float((torch.tensor(0.5051124691963196, dtype=torch.float32).to(wd) * 1000).float())
    assert vals[torch.bfloat16] == 504.0 and vals[torch.float16] == 505.0
    from omgsr_amd.diffusers_api import FluxTransformer2DModel
    cfg = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, attention_head_dim=128, joint_attention_dim=64,
               pooled_projection_dim=32, in_channels=64)
    m = FluxTransformer2DModel(**cfg)
    assert m.round_timestep_to_weight_dtype is True

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

"""Build-container drop-in check (same class of script as make_golden.py: it IMPORTS the reference, so it never runs on the GPU box).

    python tests/golden/dropin_check.py      ->  tests/golden/dropin_check.log

The reference's own `infer/omgsr_s_infer_model.py`, `infer/omgsr_f_infer_model.py` and `infer/vaehook.py` are imported UNCHANGED
with `diffusers` / `peft` pointing at `omgsr_amd.diffusers_api` (the import swap INTEGRATION.md describes), then driven as far as a
GPU-less host allows:

  1. the reference constructors `OMGSR_S_Infer(sd_path, lora_path, 273, device, dtype)` / `OMGSR_F_Infer(flux_path, lora_path,
     device, dtype)` run on CPU against an HF directory + adapter directories written to a temp dir: `from_pretrained(path,
     subfolder=)`, `PeftModel.from_pretrained` + `merge_and_unload()`, `.to(device=, dtype=)`, `.eval()`, `requires_grad_`,
     `scheduler.alphas_cumprod[t]`, `vae.config.block_out_channels` - the whole surface of SURVEY §8(b) rows 1, 5, 8;
  2. the reference's `_init_tiled_vae` installs ITS VAEHook on the product's encoder / decoder and the reference's
     `build_task_queue(net, is_decoder)` walks the product's attribute tree (rows 9-11);
  3. the reference's `forward()` is entered: it reaches the product's first kernel call (`vae.encode` -> omgsr_nchw_to_nhwc), which
     refuses a CPU tensor with OmgsrError - there is no CPU fallback to fall into.
Only a short text log is written (names, counts, shapes, the error text): no reference source, no tensors.
"""
import importlib
import io
import json
import os
import sys
import tempfile
import types
from contextlib import redirect_stdout

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SMALL_VAE = dict(block_out_channels=[32, 64, 128, 128], layers_per_block=2, norm_num_groups=32)   # the reference's task queue assumes 2 resnets per block
SMALL_UNET = dict(block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4], cross_attention_dim=128, layers_per_block=2)
SMALL_FLUX = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, attention_head_dim=128, joint_attention_dim=64,
                  pooled_projection_dim=32, in_channels=64)
SMALL_FLUX_VAE = dict(SMALL_VAE, latent_channels=16, use_quant_conv=False, use_post_quant_conv=False, scaling_factor=0.3611, shift_factor=0.1159)


def swap_imports():
    import omgsr_amd.diffusers_api as api
    d = types.ModuleType("diffusers")
    for n in ("AutoencoderKL", "UNet2DConditionModel", "DDPMScheduler", "FluxTransformer2DModel"):
        setattr(d, n, getattr(api, n))
    d.FluxPipeline = type("FluxPipeline", (), {})          # only the driver script (text encoders, out of scope) touches it
    sys.modules["diffusers"] = d
    tu = types.ModuleType("diffusers.training_utils")
    tu.free_memory = lambda: None
    sys.modules["diffusers.training_utils"] = tu
    p = types.ModuleType("peft")
    p.PeftModel = api.PeftModel
    sys.modules["peft"] = p
    tv, tvt, tvf = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")
    tv.transforms, tvt.functional = tvt, tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})


def write_adapter(path, module, targets, r=4):
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(7)
    sd = {}
    for name, m in module.named_modules():
        if any(name.endswith(t) for t in targets) and getattr(m, "weight", None) is not None and m.weight.dim() >= 2:
            w = m.weight
            a_shape = (r, w.shape[1]) + tuple(w.shape[2:])
            b_shape = (w.shape[0], r) + ((1, 1) if w.dim() == 4 else ())
            sd[f"base_model.model.{name}.lora_A.weight"] = torch.randn(a_shape, generator=g) * 0.05
            sd[f"base_model.model.{name}.lora_B.weight"] = torch.randn(b_shape, generator=g) * 0.05
    os.makedirs(path)
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump(dict(r=r, lora_alpha=r, target_modules=list(targets)), f)
    save_file(sd, os.path.join(path, "adapter_model.safetensors"))
    return len(sd) // 2


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present: this check only runs in the build container")
    sys.path.insert(0, REF)
    swap_imports()
    from omgsr_amd import _lib
    from omgsr_amd.diffusers_api import AutoencoderKL, FluxTransformer2DModel, UNet2DConditionModel
    from omgsr_amd.testing import seeded_init_
    log = []
    say = lambda s: (log.append(s), print(s))              # noqa: E731
    S = importlib.import_module("infer.omgsr_s_infer_model")
    Fm = importlib.import_module("infer.omgsr_f_infer_model")
    V = importlib.import_module("infer.vaehook")
    import infer.devices as devices
    devices.device = torch.device("cpu")
    say(f"imported the reference's infer.omgsr_s_infer_model / infer.omgsr_f_infer_model / infer.vaehook with diffusers, peft -> omgsr_amd.diffusers_api")
    torch.cuda.synchronize = lambda *a, **k: None           # the reference's forward() calls it unconditionally (SURVEY C-11)
    cpu = torch.device("cpu")
    with tempfile.TemporaryDirectory() as tmp:
        # ---------------- OMGSR-S ----------------
        sd_path, lora = os.path.join(tmp, "sd"), os.path.join(tmp, "lora_s")
        v, u = seeded_init_(AutoencoderKL(**SMALL_VAE), 1, rounded=False), seeded_init_(UNet2DConditionModel(**SMALL_UNET), 2, rounded=False)
        v.save_pretrained(sd_path, subfolder="vae")
        u.save_pretrained(sd_path, subfolder="unet", max_shard_size=8 << 20)
        os.makedirs(os.path.join(sd_path, "scheduler"))
        json.dump(dict(_class_name="DDPMScheduler", beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                       prediction_type="epsilon"), open(os.path.join(sd_path, "scheduler", "scheduler_config.json"), "w"))
        n1 = write_adapter(os.path.join(lora, "vae_encoder_lora_adapter"), v.encoder, ("conv1", "conv2", "conv_in", "conv_shortcut", "conv", "conv_out", "to_k", "to_q", "to_v", "to_out.0"))
        n2 = write_adapter(os.path.join(lora, "unet_lora_adapter"), u, ("to_k", "to_q", "to_v", "to_out.0", "conv", "conv1", "conv2", "conv_shortcut", "conv_out", "proj_in", "proj_out", "ff.net.2", "ff.net.0.proj"))
        w0 = u.down_blocks[0].resnets[0].conv1.weight.detach().clone()
        with redirect_stdout(io.StringIO()):
            pipe = S.OMGSR_S_Infer(sd_path, lora, 273, cpu, torch.float32)
        say(f"reference OMGSR_S_Infer.__init__ ran on {type(pipe.vae).__module__}.{type(pipe.vae).__name__} / {type(pipe.unet).__name__}: "
            f"alpha_t = {float(pipe.alpha_t)!r}, {n1} + {n2} LoRA targets merged, unet is a {type(pipe.unet).__name__} forwarding "
            f"config.in_channels = {pipe.unet.config.in_channels}, dtype = {pipe.unet.dtype}, training = {pipe.unet.training}")
        merged = pipe.unet.base_model.down_blocks[0].resnets[0].conv1.weight
        say(f"  merge_and_unload changed the base weights in place: max |dW| = {float((merged.detach() - w0).abs().max()):.4f}")
        with redirect_stdout(io.StringIO()):
            pipe._init_tiled_vae(encoder_tile_size=256, decoder_tile_size=64)
        say(f"reference _init_tiled_vae: vae.encoder.forward -> {type(pipe.vae.encoder.forward).__module__}.{type(pipe.vae.encoder.forward).__name__}, "
            f"original_forward kept = {hasattr(pipe.vae.encoder, 'original_forward')}")
        for net, dec in ((pipe.vae.encoder, False), (pipe.vae.decoder, True)):
            net_ = getattr(net, "base_model", net)
            q = V.build_task_queue(net_, dec)
            kinds = {}
            for k, _ in q:
                kinds[k] = kinds.get(k, 0) + 1
            say(f"reference build_task_queue({'decoder' if dec else 'encoder'}) over the product's attribute tree: {len(q)} tasks {dict(sorted(kinds.items()))}")
        try:
            with torch.no_grad(), redirect_stdout(io.StringIO()):
                pipe.forward(torch.zeros(1, 3, 128, 128), torch.zeros(1, 77, 128), 16, 8)
            say("  !! the reference forward() completed on CPU: a fallback exists")
        except _lib.OmgsrError as e:
            say(f"reference forward() reached the product's first kernel call, which refuses a CPU tensor: OmgsrError: {e}")
        # ---------------- OMGSR-F ----------------
        flux_path, lora = os.path.join(tmp, "flux"), os.path.join(tmp, "lora_f")
        fv, fl = seeded_init_(AutoencoderKL(**SMALL_FLUX_VAE), 3, rounded=False), seeded_init_(FluxTransformer2DModel(**SMALL_FLUX), 4, rounded=False)
        fv.save_pretrained(flux_path, subfolder="vae")
        fl.save_pretrained(flux_path, subfolder="transformer", max_shard_size=6 << 20)
        n3 = write_adapter(os.path.join(lora, "flux_adapter"), fl, ("to_k", "to_q", "to_v", "to_out.0", "add_k_proj", "add_q_proj", "add_v_proj", "to_add_out", "ff.net.0.proj", "ff.net.2", "proj_mlp", "proj_out", "x_embedder", "norm1.linear", "norm.linear"))
        n4 = write_adapter(os.path.join(lora, "vae_encoder_adapter"), fv.encoder, ("conv1", "conv2", "conv_in", "conv_shortcut", "conv", "conv_out"))
        with redirect_stdout(io.StringIO()):
            fp = Fm.OMGSR_F_Infer(flux_path, lora, cpu, torch.float32)
        say(f"reference OMGSR_F_Infer.__init__ ran: t_curr = {fp.t_curr!r}, t_prev = {fp.t_prev!r}, vae_scale_factor = {fp.vae_scale_factor}, "
            f"{n3} + {n4} LoRA targets merged, flux dtype = {fp.flux_transformer.dtype}")
        try:
            with torch.no_grad(), redirect_stdout(io.StringIO()):
                fp.forward(torch.zeros(1, 3, 128, 128), torch.zeros(1, 8, 64), torch.zeros(1, 32), torch.zeros(8, 3), torch.zeros(64, 3), 16, 8)
            say("  !! the reference forward() completed on CPU: a fallback exists")
        except _lib.OmgsrError as e:
            say(f"reference OMGSR_F forward() reached the product's first kernel call: OmgsrError: {e}")
    with open(os.path.join(HERE, "dropin_check.log"), "w") as f:
        f.write("\n".join(log) + "\n")


if __name__ == "__main__":
    main()

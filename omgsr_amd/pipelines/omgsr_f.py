"""OMGSR-F inference pipeline on MI355X — counterpart of infer/omgsr_f_infer_model.py::OMGSR_F_Infer
(same constructor / forward signature; encode_images, 2x2 pack, ONE FluxTransformer2DModel call at
sigma(t*=244)=0.50511, Euler step to sigma 0, unpack, un-scale, decode WITHOUT clamp; latent tiling 128/64
above 128x128 latents), executing bf16 NHWC / token-major end to end on the HIP kernels.

    z = (encode(lq).sample() - 0.1159) * 0.3611 ; x = pack(z)
    v = flux(x, sigma, guidance, pooled, prompt, ids) ; x' = x + (0 - sigma) * v
    img = decode(unpack(x') / 0.3611 + 0.1159)
"""
from __future__ import annotations

import math
import os
import time
from typing import List, Optional

import torch

from .. import ops
from ..diffusers_api import AutoencoderKL, FLUX_VAE_CONFIG, PeftModel
from ..diffusers_api.autoencoder_kl import DiagonalGaussianDistribution
from ..diffusers_api.transformer_flux import FluxTransformer2DModel
from .latent_tiling import tiled_denoise


def get_flux_setting_timesteps(n: int = 999) -> List[float]:
    """Shifted flow-matching schedule of infer/omgsr_f_infer_model.py:43-76: linspace(1, 0, n+1) in fp32,
    time_shift(mu, 1, t) = e^mu / (e^mu + (1/t - 1)) with mu linear in the packed sequence length 4096 (= 1.15)."""
    image_seq_len = (1024 // 8) * (1024 // 8) // 4
    slope = (1.15 - 0.5) / (4096 - 256)
    mu = slope * image_seq_len + (0.5 - slope * 256)
    t = torch.linspace(1, 0, n + 1)
    return (math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** 1.0)).tolist()


def prepare_latent_image_ids(height: int, width: int, device, dtype) -> torch.Tensor:
    """infer/infer_omgsr_f.py:17-28: ids[h*w, 3] = (0, row, col)."""
    yy, xx = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    return torch.stack([torch.zeros_like(yy), yy, xx], dim=-1).reshape(height * width, 3).to(device=device, dtype=dtype)


class OMGSR_F_Infer(torch.nn.Module):
    def __init__(self, flux_path: Optional[str], lora_path: Optional[str], device, weight_dtype=torch.bfloat16,
                 mid_timestep: int = 244, guidance_scale: float = 1.0, vae: Optional[AutoencoderKL] = None,
                 flux_transformer: Optional[FluxTransformer2DModel] = None, verbose: bool = False, precision_policy=None):
        super().__init__()
        ops.set_compute_dtype(weight_dtype)       # --weight_dtype picks the tier (bf16 | fp16 fast, fp32 accurate: see OMGSR_S_Infer)
        if vae is None:
            vae = AutoencoderKL.from_pretrained(flux_path, subfolder="vae")
        if flux_transformer is None:
            flux_transformer = FluxTransformer2DModel.from_pretrained(flux_path, subfolder="transformer")
        vae.requires_grad_(False)
        flux_transformer.requires_grad_(False)
        vae = vae.to(device=device, dtype=weight_dtype)
        flux_transformer = flux_transformer.to(device=device, dtype=weight_dtype)
        if lora_path:
            flux_transformer = PeftModel.from_pretrained(flux_transformer, os.path.join(lora_path, "flux_adapter"), is_trainable=False).merge_and_unload()
            vae.encoder = PeftModel.from_pretrained(vae.encoder, os.path.join(lora_path, "vae_encoder_adapter"), is_trainable=False).merge_and_unload()
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        self.guidance_scale = guidance_scale
        self.mid_timestep = mid_timestep
        self.weight_dtype = weight_dtype
        ts = get_flux_setting_timesteps()
        self.t_curr = ts[-(self.mid_timestep + 1)]
        self.t_prev = ts[-1]
        self.vae = vae.eval()
        self.flux_transformer = flux_transformer.eval()
        self.device = device
        self.verbose = verbose
        from ..precision import RangeFallback
        if weight_dtype == torch.float32:
            from ..precision import resolve
            resolve(precision_policy, vae=self.vae, flux=self.flux_transformer)
        self.range_fallback = RangeFallback(self.vae, self.flux_transformer, weight_dtype=weight_dtype)
        from .graphed import GraphCache          # hipGraph replay of forward()'s body: off by default, enable_graphs() / OMGSR_GRAPH=1
        self.graphs = GraphCache()
        self.graphs.enabled = os.environ.get("OMGSR_GRAPH", "0") == "1"
        self.range_fallback.on_mode_change = self.graphs.clear
        self._graph_params = None

    def enable_graphs(self, on: bool = True) -> None:
        """See OMGSR_S_Infer.enable_graphs."""
        self.graphs.enabled = bool(on)
        if not on:
            self.graphs.clear()

    def _weights_stamp(self) -> tuple:
        if self._graph_params is None:
            from .graphed import WeightsStamp
            self._graph_params = WeightsStamp(self.vae, self.flux_transformer)
        return self._graph_params()

    def _init_tiled_vae(self, encoder_tile_size=256, decoder_tile_size=256, fast_decoder=False, fast_encoder=False,
                        color_fix=False, vae_to_gpu=True):
        from .vaehook import VAEHook
        self.vae.encoder._tile_hook = VAEHook(self.vae.encoder, encoder_tile_size, is_decoder=False, fast_decoder=fast_decoder,
                                              fast_encoder=fast_encoder, color_fix=color_fix, to_gpu=vae_to_gpu)
        self.vae.decoder._tile_hook = VAEHook(self.vae.decoder, decoder_tile_size, is_decoder=True, fast_decoder=fast_decoder,
                                              fast_encoder=fast_encoder, color_fix=color_fix, to_gpu=vae_to_gpu)

    def _velocity_tokens(self, z_nhwc, prompt_embeds, pooled, text_ids, image_ids):
        """z [B,t,t,16] -> (packed tokens, velocity tokens) [B, t*t/4, 64]."""
        C = self.vae.config.latent_channels
        tok = ops.flux_pack(z_nhwc, C)
        B = tok.shape[0]
        # the same two tensors on every call (the DiT reads their value once per tensor: no per-image host synchronisation)
        key = (str(tok.device), B, float(self.t_curr), float(self.guidance_scale))
        if self.__dict__.get("_tg_key") != key:
            from ..nn import bump_cache_epoch
            bump_cache_epoch()          # captured hipGraphs read these two tensors by address (pipelines/graphed.py)
            self.__dict__["_tg"] = (torch.tensor([self.t_curr], device=tok.device),
                                    torch.full((B,), self.guidance_scale, device=tok.device, dtype=torch.float32))
            self.__dict__["_tg_key"] = key
        timestep, guidance = self.__dict__["_tg"]
        vel = self.flux_transformer.tokens(tok, timestep, guidance, pooled, prompt_embeds, text_ids, image_ids)
        return tok, vel

    @torch.no_grad()
    def sr_nhwc(self, lq_nhwc8, prompt_embeds, pooled, text_ids, image_ids, tile_size: int, tile_overlap: int):
        """lq [B,H,W,8] bf16 -> image NHWC [B,H,W,8] bf16 (no clamp: the reference driver clips later)."""
        sf, sh = float(self.vae.config.scaling_factor), float(self.vae.config.shift_factor)
        C = self.vae.config.latent_channels
        dt = self.t_prev - self.t_curr
        ops.timing_stage(ops.STAGE_ENCODE)
        moments = self.vae.encode_moments_nhwc(lq_nhwc8)
        post = DiagonalGaussianDistribution(moments, C, self.vae.posterior_noise, ops.stream_dtype())
        z = post.sample_nhwc(shift=sh, scale=sf)                                  # [B,h,w,16]
        _, h, w, _ = z.shape
        ops.timing_stage(ops.STAGE_DENOISE)
        if h * w <= tile_size * tile_size:
            tok, vel = self._velocity_tokens(z, prompt_embeds, pooled, text_ids, image_ids)
            tok = ops.axpby(tok, vel, 1.0 / sf, dt / sf, sh, 1.0)                  # (x + dt*v)/sf + shift, packed layout
            z1 = ops.flux_unpack(tok, h, w)
        else:
            def denoise(tile):
                _, vel = self._velocity_tokens(tile, prompt_embeds, pooled, text_ids, image_ids)
                return ops.flux_unpack(vel, tile.shape[1], tile.shape[2])
            v = tiled_denoise(z, C, tile_size, tile_overlap, denoise)
            z1 = ops.axpby(z, v, 1.0 / sf, dt / sf, sh, 1.0)
        ops.timing_stage(ops.STAGE_DECODE)
        img = self.vae.decode_nhwc(z1)
        ops.timing_stage(ops.STAGE_NONE)
        return img

    @torch.no_grad()
    def forward(self, lq_img, prompt_embeds, pooled_prompt_embeds, text_ids, latent_image_ids, tile_size, tile_overlap):
        torch.cuda.synchronize()
        start_time = time.time()
        def body(lq):
            x = ops.nchw_to_nhwc(lq.contiguous(), 8)
            img = self.sr_nhwc(x, prompt_embeds, pooled_prompt_embeds, text_ids, latent_image_ids, tile_size, tile_overlap)
            return ops.nhwc_to_nchw(img, channels=3, dtype=ops.io_dtype(lq))

        def run():
            if not self.graphs.enabled:
                return body(lq_img)
            hooks = (id(getattr(self.vae.encoder, "_tile_hook", None)), id(getattr(self.vae.decoder, "_tile_hook", None)))
            key = ("F", tile_size, tile_overlap, float(self.t_curr), float(self.guidance_scale), hooks, self._weights_stamp())
            return self.graphs.call(key, [lq_img], [prompt_embeds, pooled_prompt_embeds, text_ids, latent_image_ids, self.vae.posterior_noise], body)
        # FLUX activations are why the reference defaults to bf16: when an fp16 operand leaves the fp16 range the call runs again with
        # bf16 operands and the pipeline STAYS range-safe (precision.RangeFallback), instead of paying two passes and two re-packs per image
        pred_img = self.range_fallback.run(run, "OMGSR-F")
        t = time.time() - start_time
        if self.verbose:
            print(f"Inference time per image: {t}s")
        return pred_img, t

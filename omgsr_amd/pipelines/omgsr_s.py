"""OMGSR-S inference pipeline on MI355X — counterpart of infer/omgsr_s_infer_model.py::OMGSR_S_Infer
(same constructor / forward signature and return value, same dispatch on h*w <= tile^2, same x0
algebra and clamp), executing bf16 NHWC end to end on the HIP kernels:

    lq_img --encode--> moments --sample * 0.18215--> z --UNet(t*) [tiled 64/32]--> eps
    z0 = (z - sqrt(1 - a_t) * eps) / sqrt(a_t);  img = clamp(decode(z0 / 0.18215), -1, 1)

Batch B > 1 is the build's extension (the reference driver is batch-1, infer/infer_omgsr_s.py:92): images
are independent, prompt_embeds [1,77,1024] is broadcast.
"""
from __future__ import annotations

import math
import os
import time
from typing import Optional

import torch

from .. import ops
from ..diffusers_api import AutoencoderKL, DDPMScheduler, PeftModel, UNet2DConditionModel
from .latent_tiling import tiled_denoise


class OMGSR_S_Infer(torch.nn.Module):
    def __init__(self, sd_path: Optional[str], lora_path: Optional[str], mid_timestep: int, device, weight_dtype=torch.bfloat16,
                 vae: Optional[AutoencoderKL] = None, unet: Optional[UNet2DConditionModel] = None, verbose: bool = False,
                 precision_policy=None):
        """With `sd_path` the modules are loaded from an HF directory exactly like the reference
        (infer/omgsr_s_infer_model.py:11-25); `vae=` / `unet=` inject already-built modules instead
        (synthetic-weight benchmarks and tests — there are no checkpoints on the GPU box)."""
        super().__init__()
        # --weight_dtype picks the tier: bf16 / fp16 = that 16-bit type end to end; fp32 = the accurate tier (fp32 stream
        # tensors, fp16 MFMA operands, two-term split operands where the precision policy says so)
        ops.set_compute_dtype(weight_dtype)
        self.mid_timestep = mid_timestep
        self.verbose = verbose
        if vae is None:
            vae = AutoencoderKL.from_pretrained(sd_path, subfolder="vae")
        if unet is None:
            unet = UNet2DConditionModel.from_pretrained(sd_path, subfolder="unet")
        self.scheduler = DDPMScheduler.from_pretrained(sd_path, subfolder="scheduler") if sd_path else DDPMScheduler()
        self.alpha_t = self.scheduler.alphas_cumprod[mid_timestep]
        if lora_path:
            vae.encoder = PeftModel.from_pretrained(vae.encoder, os.path.join(lora_path, "vae_encoder_lora_adapter"))
            unet = PeftModel.from_pretrained(unet, os.path.join(lora_path, "unet_lora_adapter"))
            vae.encoder = vae.encoder.merge_and_unload()
            unet = unet.merge_and_unload()
        self.vae = vae.to(device=device, dtype=weight_dtype).eval()
        self.unet = unet.to(device=device, dtype=weight_dtype).eval()
        self.device = device
        from ..precision import RangeFallback
        if weight_dtype == torch.float32:       # which layers carry two-term split operands / weights (omgsr_amd/precision.py)
            from ..precision import resolve
            resolve(precision_policy, vae=self.vae, unet=self.unet)
        self.range_fallback = RangeFallback(self.vae, self.unet, weight_dtype=weight_dtype)
        # hipGraph replay of forward()'s body (pipelines/graphed.py): off by default, enable_graphs() / OMGSR_GRAPH=1
        from .graphed import GraphCache
        self.graphs = GraphCache()
        self.graphs.enabled = os.environ.get("OMGSR_GRAPH", "0") == "1"
        self.range_fallback.on_mode_change = self.graphs.clear
        self._graph_params = None

    def enable_graphs(self, on: bool = True) -> None:
        """Capture forward()'s launches into a hipGraph per (input shape, tile geometry, prompt tensor, tier) and replay it: one host
        call per image instead of ~1 000 - 2 200 kernel launches issued from Python (the reference's batch-1 operating point is
        launch-bound otherwise). Call `self.graphs.clear()` after changing weights in place."""
        self.graphs.enabled = bool(on)
        if not on:
            self.graphs.clear()

    def _weights_stamp(self) -> tuple:
        if self._graph_params is None:
            from .graphed import WeightsStamp
            self._graph_params = WeightsStamp(self.vae, self.unet)
        return self._graph_params()

    def _init_tiled_vae(self, encoder_tile_size=256, decoder_tile_size=256, fast_decoder=False, fast_encoder=False,
                        color_fix=False, vae_to_gpu=True):
        from .vaehook import VAEHook
        self.vae.encoder._tile_hook = VAEHook(self.vae.encoder, encoder_tile_size, is_decoder=False, fast_decoder=fast_decoder,
                                              fast_encoder=fast_encoder, color_fix=color_fix, to_gpu=vae_to_gpu)
        self.vae.decoder._tile_hook = VAEHook(self.vae.decoder, decoder_tile_size, is_decoder=True, fast_decoder=fast_decoder,
                                              fast_encoder=fast_encoder, color_fix=color_fix, to_gpu=vae_to_gpu)

    # ---- NHWC hot path ---------------------------------------------------------------------
    @torch.no_grad()
    def sr_nhwc(self, lq_nhwc8: torch.Tensor, prompt_embeds: torch.Tensor, tile_size: int, tile_overlap: int) -> torch.Tensor:
        """lq [B,H,W,8] stream tensor (RGB + zero pad) -> image NHWC [B,H,W,8] stream tensor, UNCLAMPED."""
        sf = float(self.vae.config.scaling_factor)
        ops.timing_stage(ops.STAGE_ENCODE)
        moments = self.vae.encode_moments_nhwc(lq_nhwc8)
        post = self.vae_posterior(moments)
        z = post.sample_nhwc(shift=0.0, scale=sf)                                   # [B,h,w,8]
        _, h, w, _ = z.shape
        ops.timing_stage(ops.STAGE_DENOISE)
        if h * w <= tile_size * tile_size:
            if self.verbose:
                print("[Tiled Latent]: the input size is tiny and unnecessary to tile.")
            eps = self.unet.nhwc(z, self.mid_timestep, prompt_embeds)
        else:
            if self.verbose:
                print(f"[Tiled Latent]: the input size is {lq_nhwc8.shape[1]}x{lq_nhwc8.shape[2]}, need to tiled")
            eps = tiled_denoise(z, self.unet.config.in_channels, tile_size, tile_overlap,
                                lambda t: self.unet.nhwc(t, self.mid_timestep, prompt_embeds))
        a = float(self.alpha_t)
        s1, s2 = math.sqrt(1.0 - a), math.sqrt(a)
        # (z - s1*eps) / s2 / scaling_factor  in one fused pass
        z0 = ops.axpby(z, eps, 1.0, -s1, 0.0, 1.0 / (s2 * sf))
        ops.timing_stage(ops.STAGE_DECODE)
        img = self.vae.decode_nhwc(z0)
        ops.timing_stage(ops.STAGE_NONE)
        return img

    def vae_posterior(self, moments):
        from ..diffusers_api.autoencoder_kl import DiagonalGaussianDistribution
        return DiagonalGaussianDistribution(moments, self.vae.config.latent_channels, self.vae.posterior_noise, ops.stream_dtype())

    # ---- reference API ---------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, lq_img: torch.Tensor, prompt_embeds: torch.Tensor, tile_size: int, tile_overlap: int):
        torch.cuda.synchronize()
        start_time = time.time()
        def body(lq):
            x = ops.nchw_to_nhwc(lq.contiguous(), 8)
            img = self.sr_nhwc(x, prompt_embeds, tile_size, tile_overlap)
            return ops.nhwc_to_nchw(img, channels=3, dtype=ops.io_dtype(lq), clamp=(-1.0, 1.0))

        def run():
            if not self.graphs.enabled:
                return body(lq_img)
            hooks = (id(getattr(self.vae.encoder, "_tile_hook", None)), id(getattr(self.vae.decoder, "_tile_hook", None)))
            key = ("S", tile_size, tile_overlap, self.mid_timestep, hooks, self._weights_stamp())
            return self.graphs.call(key, [lq_img], [prompt_embeds, self.vae.posterior_noise], body)
        # an fp16 operand that leaves the fp16 range somewhere in this call (the stores saturate at +-65504) makes the call run again
        # range-safe, and the pipeline stays that way (precision.RangeFallback: `self.range_fallback.count / .sticky`)
        pred_img = self.range_fallback.run(run, "OMGSR-S")
        t = time.time() - start_time
        if self.verbose:
            print(f"Inference time per image: {t}s")
        return pred_img, t

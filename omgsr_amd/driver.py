"""The reference driver's per-image loop body, whole, on the device (SURVEY §8 row a17 + f1 + f2).

infer/infer_omgsr_s.py:69-107 does, per image: PIL open -> (small inputs: bicubic up to process_size // upscale, remembered as
`resize_flag`) -> bicubic x upscale -> LANCZOS snap to multiples of 8 -> to_tensor * 2 - 1 -> net_sr(lq, prompt, process_size // 8,
process_size // 16) -> * 0.5 + 0.5, clip, ToPILImage -> AdaIN / wavelet colour fix against the resized input -> (resize_flag: bicubic
back to upscale x the ORIGINAL size) -> save. Here every step between the decoded uint8 image and the uint8 result is a kernel:
omgsr_amd.preprocess (Pillow's 8-bit resampler bit for bit), the pipeline's NHWC path, omgsr_amd.colorfix.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import preprocess
from .colorfix import super_resolve_u8


@torch.no_grad()
def sr_image_u8(pipe, image_u8: torch.Tensor, prompt_embeds: torch.Tensor, *extra_model_args, process_size: int = 512,
                upscale: int = 4, align_method: Optional[str] = "adain") -> torch.Tensor:
    """image_u8: uint8 [B,H,W,3] on the device (PIL memory order, one size per batch) -> uint8 [B, upscale*H', upscale*W', 3].
    `pipe` is an OMGSR_S_Infer (extra_model_args empty) or OMGSR_F_Infer (pooled, text_ids, latent_image_ids); the latent tile
    size / overlap are the driver's `process_size // 8` and half of it (infer/infer_omgsr_s.py:87-88)."""
    _, h, w, _ = image_u8.shape
    resize_flag = w < process_size // upscale or h < process_size // upscale       # infer/infer_omgsr_s.py:76-79
    lq = preprocess.preprocess_u8(image_u8, process_size, upscale)
    tile_size = process_size // 8
    out = super_resolve_u8(pipe, lq, prompt_embeds, *extra_model_args, tile_size, tile_size // 2, align_method=align_method)
    if resize_flag:                                                                # :104-105 (Image.resize default: BICUBIC)
        out = preprocess.resize_u8(out, (int(upscale * w), int(upscale * h)), preprocess.BICUBIC)
    return out

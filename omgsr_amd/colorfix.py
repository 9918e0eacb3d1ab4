"""SURVEY §8(f) f1 — GPU counterpart of the reference driver's post-process (`infer/infer_omgsr_s.py:96-103`,
`infer/wavelet_color_fix.py`): model output -> uint8 image, optionally colour-aligned to the upscaled LQ input with
AdaIN (`--align_method adain`, the default) or the wavelet low-frequency swap (`--align_method wavelet`).

    img_u8 = color_fix(sr_nhwc, source_u8, "adain")      # [B,H,W,3] uint8 on the device, PIL memory order

Same names / argument meaning as the reference's `adain_color_fix(target, source)` / `wavelet_color_fix(target, source)`
at the tensor level; the PIL round trip (ToPILImage truncation, ToTensor / 255) is part of the kernels."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, ops
from ._lib import check

METHODS = {"nofix": 0, "none": 0, None: 0, "adain": 1, "wavelet": 2}


def color_fix(sr_nhwc: torch.Tensor, source_u8: Optional[torch.Tensor], method: Optional[str] = "adain") -> torch.Tensor:
    """sr_nhwc [B,H,W,C>=3] (compute dtype, [-1,1] unclamped) + source_u8 [B,H,W,3] uint8 -> uint8 [B,H,W,3]."""
    if method not in METHODS:
        raise ValueError(f"align_method must be one of {sorted(k for k in METHODS if k)}; got {method!r}")
    m = METHODS[method]
    sr_el = ops._el(sr_nhwc, "sr_nhwc")
    B, H, W, ld = sr_nhwc.shape
    if m:
        if source_u8 is None:
            raise ValueError("the colour fix needs the source (upscaled LQ) image")
        ops._req(source_u8, torch.uint8, "source_u8")
        if tuple(source_u8.shape) != (B, H, W, 3):
            raise ValueError(f"source_u8 must be [B,H,W,3] = {(B, H, W, 3)}, got {tuple(source_u8.shape)}")
    lib = _lib.load()
    out = torch.empty((B, H, W, 3), device=sr_nhwc.device, dtype=torch.uint8)
    need = lib.omgsr_colorfix_workspace_bytes(B, H, W, m)
    ws = torch.empty(max(need, 8), device=sr_nhwc.device, dtype=torch.uint8) if m else None
    check(lib.omgsr_colorfix(sr_nhwc.data_ptr(), ld, ops._ptr(source_u8) if m else None, out.data_ptr(), ops._ptr(ws),
                             B, H, W, m, sr_el, ops._stream()), "omgsr_colorfix")
    return out


def adain_color_fix(target_nhwc: torch.Tensor, source_u8: torch.Tensor) -> torch.Tensor:
    return color_fix(target_nhwc, source_u8, "adain")


def wavelet_color_fix(target_nhwc: torch.Tensor, source_u8: torch.Tensor) -> torch.Tensor:
    return color_fix(target_nhwc, source_u8, "wavelet")


def image_to_model_input(image_u8: torch.Tensor) -> torch.Tensor:
    """uint8 [B,H,W,3] -> NHWC stream tensor [B,H,W,8]: `F.to_tensor(img).to(weight_dtype) * 2 - 1` (infer/infer_omgsr_s.py:92)."""
    ops._req(image_u8, torch.uint8, "image_u8")
    B, H, W, c = image_u8.shape
    if c != 3:
        raise ValueError("image_u8 must be [B,H,W,3]")
    out = torch.empty((B, H, W, 8), device=image_u8.device, dtype=ops.stream_dtype())
    check(_lib.load().omgsr_image_to_model_input(image_u8.data_ptr(), out.data_ptr(), B, H, W,
                                                 ops.EL_F32 if out.dtype == torch.float32 else ops.EL_16, ops._stream()), "omgsr_image_to_model_input")
    return out


@torch.no_grad()
def super_resolve_u8(pipe, image_u8: torch.Tensor, *model_args, align_method: Optional[str] = "adain") -> torch.Tensor:
    """The reference driver's per-image loop body (infer/infer_omgsr_s.py:90-103) without leaving the device:
    uint8 upscaled-LQ images [B,H,W,3] -> model -> uint8 colour-fixed SR images [B,H,W,3].
    `pipe` is an OMGSR_S_Infer / OMGSR_F_Infer; model_args are what its sr_nhwc takes after the image."""
    x = image_to_model_input(image_u8)
    img = pipe.sr_nhwc(x, *model_args)
    return color_fix(img.contiguous(), image_u8, align_method)

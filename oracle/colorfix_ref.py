"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference driver's post-processing:
uint8 conversion + AdaIN / wavelet colour fix. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this; the product path (omgsr_amd/colorfix.py -> libomgsr_hip.so) never does.

Follows: infer/infer_omgsr_s.py:96-103 (x*0.5+0.5 in the weight dtype, clip, float, ToPILImage),
infer/wavelet_color_fix.py:12-25 (adain_color_fix), :28-41 (wavelet_color_fix), :44-57 (calc_mean_std: UNBIASED var + 1e-5),
:60-74 (adaptive_instance_normalization), :77-96 (wavelet_blur: 3x3 [1 2 1]^2/16, dilation = radius, replicate pad),
:99-111 (wavelet_decomposition, 5 levels, radius 2^i), :114-125 (wavelet_reconstruction).
torchvision's ToTensor / ToPILImage (absent here, pinned torchvision==0.20.1) are restated from their documented
behaviour: uint8 -> float32 / 255;  float [0,1] -> mul(255).byte() (TRUNCATION).
PINNED: tests/golden/colorfix.npz holds outputs of the reference's own tensor-level functions
(tests/golden/make_golden_colorfix.py); tests/test_colorfix_golden.py checks this file against them.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def to_pil_u8(x01: torch.Tensor) -> torch.Tensor:
    """ToPILImage on a float tensor in [0, 1]: mul(255).byte() — truncation toward zero."""
    return x01.mul(255).to(torch.uint8)


def to_tensor_f32(u8: torch.Tensor) -> torch.Tensor:
    """ToTensor on a uint8 image: float32 / 255."""
    return u8.to(torch.float32).div(255)


def model_output_to_u8(out_m11: torch.Tensor) -> torch.Tensor:
    """infer/infer_omgsr_s.py:96-98: output*0.5+0.5 IN THE MODEL DTYPE, clip, .float(), ToPILImage."""
    y = out_m11 * 0.5 + 0.5
    return to_pil_u8(torch.clip(y, 0, 1).float())


def lq_to_u8(lq_m11: torch.Tensor) -> torch.Tensor:
    """Inverse of `to_tensor(img) * 2 - 1` (infer/infer_omgsr_s.py:92) for an fp32 tensor that came from a uint8 image."""
    return torch.round((lq_m11.float() + 1.0) * 127.5).clamp(0, 255).to(torch.uint8)


def calc_mean_std(feat: torch.Tensor, eps: float = 1e-5):
    b, c = feat.shape[:2]
    var = feat.reshape(b, c, -1).var(dim=2) + eps          # unbiased
    std = var.sqrt().reshape(b, c, 1, 1)
    mean = feat.reshape(b, c, -1).mean(dim=2).reshape(b, c, 1, 1)
    return mean, std


def adaptive_instance_normalization(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    sm, ss = calc_mean_std(style)
    cm, cs = calc_mean_std(content)
    return (content - cm) / cs * ss + sm


def wavelet_blur(image: torch.Tensor, radius: int) -> torch.Tensor:
    k = torch.tensor([[0.0625, 0.125, 0.0625], [0.125, 0.25, 0.125], [0.0625, 0.125, 0.0625]], dtype=image.dtype)
    k = k[None, None].repeat(3, 1, 1, 1)
    image = F.pad(image, (radius, radius, radius, radius), mode="replicate")
    return F.conv2d(image, k, groups=3, dilation=radius)


def wavelet_decomposition(image: torch.Tensor, levels: int = 5):
    high = torch.zeros_like(image)
    low = image
    for i in range(levels):
        low = wavelet_blur(image, 2 ** i)
        high = high + (image - low)
        image = low
    return high, low


def wavelet_reconstruction(content: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
    ch, _ = wavelet_decomposition(content)
    _, sl = wavelet_decomposition(style)
    return ch + sl


def adain_color_fix_u8(target_u8: torch.Tensor, source_u8: torch.Tensor) -> torch.Tensor:
    """[B,3,H,W] uint8 x2 -> uint8 (one image at a time in the reference; per-image statistics either way)."""
    r = adaptive_instance_normalization(to_tensor_f32(target_u8), to_tensor_f32(source_u8))
    return to_pil_u8(r.clamp(0.0, 1.0))


def wavelet_color_fix_u8(target_u8: torch.Tensor, source_u8: torch.Tensor) -> torch.Tensor:
    r = wavelet_reconstruction(to_tensor_f32(target_u8), to_tensor_f32(source_u8))
    return to_pil_u8(r.clamp(0.0, 1.0))

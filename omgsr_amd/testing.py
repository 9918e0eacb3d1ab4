"""Synthetic weights / inputs / metrics shared by tests, bench.py and __graft_entry__.smoke().

There are no checkpoints on the GPU box (and no network), so every model runs with seeded random
weights at the real architecture shapes. All values are bf16-representable so the
fp32 CPU oracle and the bf16 HIP path start from bit-identical parameters.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


@torch.no_grad()
def seeded_init_(model: nn.Module, seed: int = 0, device=None, rounded: bool = True) -> nn.Module:
    """Fan-in scaled normal weights, small random biases, norm affines near (1, 0). Deterministic per parameter NAME (not
    traversal order), generated on CPU. rounded=True: values rounded to bf16-representable ones (every tier then starts from
    bit-identical parameters); rounded=False: full fp32 mantissas - what a real checkpoint looks like after an fp32 LoRA merge
    (infer/omgsr_s_infer_model.py:16-23), the case the accurate tier's weight-side split exists for."""
    for name, p in model.named_parameters():
        g = torch.Generator().manual_seed((hash_name(name) + seed) & 0x7FFFFFFF)
        shape = tuple(p.shape)
        if p.dim() >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            v = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif name.endswith("weight"):       # norm scale / RMSNorm weight
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:                               # biases
            v = 0.05 * torch.randn(shape, generator=g)
        if rounded:
            v = v.to(torch.bfloat16)
        p.copy_(v.to(p.dtype).to(p.device))
    return model


@torch.no_grad()
def seeded_init_device_(model: nn.Module, seed: int = 0) -> nn.Module:
    """Same recipe as seeded_init_ but generated ON the parameter's device (FLUX.1-dev has 11.9 B parameters:
    a CPU randn of that size takes minutes). Values differ from the CPU recipe; use only where no CPU twin
    of the model is needed (throughput benchmarks)."""
    for name, p in model.named_parameters():
        g = torch.Generator(device=p.device).manual_seed((hash_name(name) + seed) & 0x7FFFFFFF)
        shape = tuple(p.shape)
        if p.dim() >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            p.copy_(torch.randn(shape, generator=g, device=p.device, dtype=torch.float32).mul_(1.0 / math.sqrt(fan_in)))
        elif name.endswith("weight"):
            p.copy_(torch.randn(shape, generator=g, device=p.device).mul_(0.1).add_(1.0))
        else:
            p.copy_(torch.randn(shape, generator=g, device=p.device).mul_(0.05))
    return model


def hash_name(name: str) -> int:
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def synthetic_lq(batch: int, height: int, width: int, seed: int = 1234) -> torch.Tensor:
    """[B,3,H,W] fp32 in [-1,1]: low-passed random field upsampled x4 (mimics the x4-upscaled LQ image the
    reference feeds the model, infer/infer_omgsr_s.py:81-84)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(batch, 3, height // 4, width // 4, generator=g)
    for _ in range(3):
        u = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(u, (1, 1, 1, 1), mode="replicate"), 3, stride=1)
    x = torch.nn.functional.interpolate(u, size=(height, width), mode="bicubic", align_corners=False)
    x = (x - x.amin(dim=(1, 2, 3), keepdim=True)) / (x.amax(dim=(1, 2, 3), keepdim=True) - x.amin(dim=(1, 2, 3), keepdim=True))
    return (x * 2 - 1).to(torch.bfloat16).float()


def rel_l2(got: torch.Tensor, ref: torch.Tensor) -> float:
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-20)).item()


def psnr(got: torch.Tensor, ref: torch.Tensor, peak: float = 2.0) -> float:
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    mse = (got - ref).pow(2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)

"""Leaf modules of the MI355X path: torch.nn parameter containers (so diffusers state dicts, `.to()`
and PEFT-style weight merging work unchanged) whose compute runs through the HIP kernels.

Two ways in:
  * `.nhwc(x, ...)`  — fast path used by the model executors: bf16 channels-last in and out
  * `.forward(x)`    — diffusers/torch calling convention (NCHW for convs/GroupNorm) so code that
                       duck-types the modules op by op (the reference's infer/vaehook.py:248-276)
                       still works; it converts layout around the same kernels.
Packed (KRSC bf16) weights are built lazily and rebuilt if the parameter changes (LoRA merge, .to()).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops


def _key(*tensors) -> tuple:
    return (ops.act_dtype(),) + tuple((t.data_ptr(), t._version, str(t.device), t.dtype) if t is not None else None for t in tensors)


class _Packed:
    """Mixin: lazily packed weights keyed on the parameters' identity/version."""

    def _packed(self, builder, *tensors):
        k = _key(*tensors)
        if getattr(self, "_pk_key", None) != k:
            self._pk = builder()
            self._pk_key = k
        return self._pk


class Conv2d(nn.Conv2d, _Packed):
    def packed(self) -> ops.PackedWeight:
        # logical Cout widened to a multiple of 8 (zero rows): 3/4-channel heads write 16-byte NHWC rows
        return self._packed(lambda: ops.pack_conv_weight(self.weight, self.bias, cout_multiple=8), self.weight, self.bias)

    def nhwc(self, x, *, pad=None, upsample=False, act=ops.ACT_NONE, residual=None, bias_override=None, stride=None, gn_groups=0):
        """x [N,H,W,Cin8] -> [N,Ho,Wo,Cout8]; channels beyond out_channels are exact zeros."""
        pw = self.packed()
        if bias_override is not None:
            pw = ops.PackedWeight(pw.w, bias_override, pw.cout, pw.cin, pw.R, pw.S, w_cm=pw.w_cm)
        p = self.padding[0] if pad is None else pad
        return ops.conv2d(x, pw, stride=stride or self.stride[0], pad=p, upsample=upsample, act=act, residual=residual,
                          gn_groups=gn_groups)

    def forward(self, x):  # NCHW compat
        y = self.nhwc(ops.nchw_to_nhwc(x.contiguous(), ops._round_up(self.in_channels, 8)))
        return ops.nhwc_to_nchw(y, channels=self.out_channels,
                                dtype=ops.io_dtype(x))


class Linear(nn.Linear, _Packed):
    def packed(self) -> ops.PackedWeight:
        return self._packed(lambda: ops.pack_linear_weight(self.weight, self.bias), self.weight, self.bias)

    def nhwc(self, x, *, act=ops.ACT_NONE, residual=None, gate=None, out_dtype=ops.OUT_BF16, gn_groups=0):
        return ops.linear(x, self.packed(), act=act, residual=residual, gate=gate, out_dtype=out_dtype, gn_groups=gn_groups)

    def forward(self, x):
        y = self.nhwc(x.to(ops.act_dtype()).contiguous())
        return y.to(x.dtype)


class GroupNorm(nn.GroupNorm, _Packed):
    def _affine(self):
        return self._packed(lambda: (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous()),
                            self.weight, self.bias)

    def nhwc(self, x, act=ops.ACT_NONE):
        g, b = self._affine()
        return ops.group_norm(x, g, b, self.num_groups, self.eps, act)

    def stats(self, x):
        return ops.group_norm_stats(x, self.num_groups, self.eps)

    def apply_stats(self, x, mean, rstd, act=ops.ACT_NONE):
        """Normalise with externally supplied per-(n, group) statistics (tiled VAE)."""
        g, b = self._affine()
        if mean.shape[0] != x.shape[0]:       # tile-major rows sharing their image's statistics
            return ops.group_norm_apply_shared(x, mean, rstd, g, b, self.num_groups, act)
        return ops.group_norm_apply(x, mean, rstd, g, b, self.num_groups, act)

    def forward(self, x):  # NCHW (or [B, C, L]) compat
        shp = x.shape
        x4 = x.reshape(shp[0], shp[1], -1, 1) if x.dim() == 3 else x
        y = self.nhwc(ops.nchw_to_nhwc(x4.contiguous()))
        return ops.nhwc_to_nchw(y, dtype=x.dtype).reshape(shp)


class LayerNorm(nn.LayerNorm, _Packed):
    def _affine(self):
        if not self.elementwise_affine:
            return None, None
        return self._packed(lambda: (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous()),
                            self.weight, self.bias)

    def nhwc(self, x, a=None, b=None):
        if a is None and b is None:
            a, b = self._affine()
        return ops.layer_norm(x, a, b, self.eps)

    def forward(self, x):
        return self.nhwc(x.to(ops.act_dtype()).contiguous()).to(x.dtype)


class RMSNormWeight(nn.Module):
    """Parameter holder for diffusers' RMSNorm (`weight` [dim]); applied inside omgsr_rmsnorm_rope."""

    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def w32(self):
        return self.weight.detach().float().contiguous()

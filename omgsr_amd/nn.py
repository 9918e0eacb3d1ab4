"""Leaf modules of the MI355X path: torch.nn parameter containers (so diffusers state dicts, `.to()`
and PEFT-style weight merging work unchanged) whose compute runs through the HIP kernels.

Two ways in:
  * `.nhwc(x, ...)`  — fast path used by the model executors: channels-last; stream tensors in and out by default
                       (ops.py: the compute type in the fast tiers, fp32 in the accurate tier), MFMA operands in between
  * `.forward(x)`    — diffusers/torch calling convention (NCHW for convs/GroupNorm) so code that
                       duck-types the modules op by op (the reference's infer/vaehook.py:248-276)
                       still works; it converts layout around the same kernels.
Packed (KRSC bf16) weights are built lazily and rebuilt if the parameter changes (LoRA merge, .to()).
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import torch
import torch.nn as nn

from . import ops


def _key(*tensors) -> tuple:
    """Identity of PARAMETER tensors (held alive by their module, so an address names one tensor) under the current tier.
    Not for per-call inputs: see input_key()."""
    return ops.mode_key() + tuple((t.data_ptr(), t._version, str(t.device), t.dtype, tuple(t.shape)) if torch.is_tensor(t) else t for t in tensors)


# Every rebuild of a cached device value (packed weights, folded constants, cross-attention K / V^T, modulation / rope tables) bumps this
# counter. A captured hipGraph bakes the ADDRESSES of those values into its launches, and the one-slot caches free the old value when they
# rebuild: pipelines/graphed.py stamps each graph with the epoch it was captured under and drops it when any cache has rebuilt since
# (prompt A -> prompt B -> prompt A would otherwise replay graph A against freed or re-used K / V^T memory).
_CACHE_EPOCH = 0


def cache_epoch() -> int:
    return _CACHE_EPOCH


def bump_cache_epoch() -> None:
    global _CACHE_EPOCH
    _CACHE_EPOCH += 1


# Rebuild log (pipelines/graphed.py): while a hipGraph capture is open, every cache that rebuilds registers how to INVALIDATE itself. A value built
# under capture was never computed (its kernels were recorded, not run); if the capture is then discarded, the slots are emptied so that the eager
# re-run rebuilds them for real instead of reading uninitialised memory.
_REBUILD_LOG = None


def begin_rebuild_log() -> None:
    global _REBUILD_LOG
    _REBUILD_LOG = []


def end_rebuild_log() -> list:
    global _REBUILD_LOG
    log, _REBUILD_LOG = _REBUILD_LOG or [], None
    return log


def _log_rebuild(invalidate) -> None:
    if _REBUILD_LOG is not None:
        _REBUILD_LOG.append(invalidate)


class InputCache:
    """One-slot cache keyed on per-call INPUT tensors (prompt embeddings, ids, pooled projections). The caching allocator
    reuses addresses, so an address is not an identity: the slot keeps a strong reference to every keyed tensor and
    compares with `is` + `_version` (+ the weights' parameter key)."""

    def __init__(self):
        self._inputs, self._versions, self._wkey, self._value = None, None, None, None

    def prime(self, inputs: tuple, wkey: tuple, value) -> None:
        """Install a value computed elsewhere (omgsr_amd.constants: the serialised constant cache)."""
        self._value, self._inputs, self._wkey = value, tuple(inputs), wkey
        self._versions = tuple(None if t is None else t._version for t in inputs)
        bump_cache_epoch()

    def invalidate(self) -> None:
        self._inputs, self._versions, self._wkey, self._value = None, None, None, None
        bump_cache_epoch()

    def get(self, inputs: tuple, wkey: tuple, builder):
        same = (self._inputs is not None and len(self._inputs) == len(inputs) and self._wkey == wkey and
                all(a is b for a, b in zip(self._inputs, inputs)) and
                self._versions == tuple(None if t is None else t._version for t in inputs))
        if not same:
            bump_cache_epoch()
            _log_rebuild(self.invalidate)
            self._value = builder()
            self._inputs, self._wkey = tuple(inputs), wkey
            self._versions = tuple(None if t is None else t._version for t in inputs)
        return self._value


class _Packed:
    """Mixin: lazily packed weights keyed on the parameters' identity/version."""

    def _drop_packed(self) -> None:
        self._pk, self._pk_key = None, None
        bump_cache_epoch()

    def _packed(self, builder, *tensors):
        k = _key(*tensors)
        if getattr(self, "_pk_key", None) != k:
            bump_cache_epoch()
            _log_rebuild(self._drop_packed)
            self._pk = builder()
            self._pk_key = k
        return self._pk


class _Operand:
    """`op_split`: 1, or 2 when this layer's INPUT arrives as a two-term split operand (accurate tier, set per layer by a
    precision policy: omgsr_amd.precision). The producer of the input (norm / cast / GEMM epilogue) asks the consumer."""
    op_split = 1
    # `w_split`: 1, or 2 when this layer's WEIGHT is carried as the two-term split w_hi + w_lo (accurate tier, fp32 checkpoints whose
    # values are not 16-bit representable; precision.set_weight_split). Invisible to producers: the operand row is unchanged, the
    # packed weight gains a [w_lo] K segment that wraps over the operand's (hi) half (ops.pack_conv_weight).
    w_split = 1
    # accurate tier: this layer's OUTPUT is only ever normalised and fed to the next GEMM (a ResnetBlock's conv1): keep it in the
    # 16-bit compute type instead of the fp32 stream type where the policy says the rounding is affordable (precision.py)
    out_inner16 = False

    def in_split(self) -> int:
        return self.op_split if ops.precise() else 1

    def in_wsplit(self) -> int:
        return self.w_split if ops.precise() else 1


class Conv2d(nn.Conv2d, _Packed, _Operand):
    # set by Upsample2D: this 3x3 conv is applied to a nearest-2x upsampled map, so its packed weight also carries the four
    # phase-summed 2 x 2 kernels (ops.pack_conv_weight upsample_phases) and runs with 4 / 9 of the MFMA work
    phase_upsample = False

    def packed(self) -> ops.PackedWeight:
        # logical Cout widened to a multiple of 8 (zero rows): 3/4-channel heads write 16-byte NHWC rows
        sp, wsp, ph = self.in_split(), self.in_wsplit(), self.phase_upsample
        return self._packed(lambda: ops.pack_conv_weight(self.weight, self.bias, cout_multiple=8, split=sp, w_split=wsp, upsample_phases=ph),
                            self.weight, self.bias, sp, wsp, ph)

    def nhwc(self, x, *, pad=None, upsample=False, act=ops.ACT_NONE, residual=None, bias_override=None, stride=None, gn_groups=0,
             out_dtype=ops.OUT_STREAM, out_split=1, gn=None):
        """x [N,H,W,Cin8] (operand, or a stream tensor that is cast / split here) -> [N,Ho,Wo,Cout8]; channels beyond
        out_channels are exact zeros. gn (ops.GnSpec): x is the stream tensor a GroupNorm reads; the norm runs as this conv's patch
        producer where the kernel can, as the separate apply pass otherwise (ops.conv2d)."""
        pw = self.packed()
        if bias_override is not None:
            pw = dataclasses.replace(pw, bias=bias_override)
        p = self.padding[0] if pad is None else pad
        if out_dtype == ops.OUT_STREAM and self.out_inner16 and ops.precise():
            out_dtype = ops.OUT_BF16
        return ops.conv2d(x, pw, stride=stride or self.stride[0], pad=p, upsample=upsample, act=act, residual=residual,
                          gn_groups=gn_groups, out_dtype=out_dtype, out_split=out_split, gn=gn)

    def nhwc_multi(self, xs, *, pad=None, upsample=False, act=ops.ACT_NONE, residuals=None, stride=None, gn_groups=0,
                   out_dtype=ops.OUT_STREAM, out_split=1, gn=None):
        """nhwc() of several inputs (the tile-shape groups of a tiled-VAE layer) in one launch where the kernel allows (ops.conv2d_multi)."""
        pw = self.packed()
        p = self.padding[0] if pad is None else pad
        if out_dtype == ops.OUT_STREAM and self.out_inner16 and ops.precise():
            out_dtype = ops.OUT_BF16
        return ops.conv2d_multi(list(xs), pw, stride=stride or self.stride[0], pad=p, upsample=upsample, act=act, residuals=residuals,
                                gn_groups=gn_groups, out_dtype=out_dtype, out_split=out_split, gn=gn)

    def forward(self, x):  # NCHW compat
        y = self.nhwc(ops.nchw_to_nhwc(x.contiguous(), ops._round_up(self.in_channels, 8)))
        return ops.nhwc_to_nchw(y, channels=self.out_channels,
                                dtype=ops.io_dtype(x))


class Linear(nn.Linear, _Packed, _Operand):
    def packed(self) -> ops.PackedWeight:
        sp, wsp = self.in_split(), self.in_wsplit()
        return self._packed(lambda: ops.pack_linear_weight(self.weight, self.bias, split=sp, w_split=wsp), self.weight, self.bias, sp, wsp)

    def nhwc(self, x, *, act=ops.ACT_NONE, residual=None, gate=None, out_dtype=ops.OUT_STREAM, gn_groups=0, out_split=1):
        return ops.linear(x, self.packed(), act=act, residual=residual, gate=gate, out_dtype=out_dtype, gn_groups=gn_groups,
                          out_split=out_split)

    def forward(self, x):
        y = self.nhwc(x.to(ops.stream_dtype()).contiguous())
        return y.to(x.dtype)


class GroupNorm(nn.GroupNorm, _Packed):
    def _affine(self):
        return self._packed(lambda: (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous()),
                            self.weight, self.bias)

    def nhwc(self, x, act=ops.ACT_NONE, split=1, also_cast=0):
        """Stream tensor -> normalised (+SiLU) MFMA operand; split 2: as the two-term split its consumer asked for.
        also_cast 1 | 2: returns (operand, x itself as a plain / split operand) from the same pass over x."""
        g, b = self._affine()
        return ops.group_norm(x, g, b, self.num_groups, self.eps, act, split=split, also_cast=also_cast)

    def stats(self, x):
        return ops.group_norm_stats(x, self.num_groups, self.eps)

    def spec(self, x, act=ops.ACT_NONE) -> "ops.GnSpec":
        """This norm (statistics of x included) as something a conv can run as its patch producer: conv.nhwc(x, gn=norm.spec(x, SILU))."""
        mean, rstd, _ = self.stats(x)
        return self.spec_stats(mean, rstd, act)

    def spec_stats(self, mean, rstd, act=ops.ACT_NONE) -> "ops.GnSpec":
        """The same with externally supplied per-(image, group) statistics (tiled VAE)."""
        g, b = self._affine()
        return ops.GnSpec(mean, rstd, g, b, self.num_groups, act)

    def apply_stats(self, x, mean, rstd, act=ops.ACT_NONE, split=1, also_cast=0):
        """Normalise with externally supplied per-(n, group) statistics (tiled VAE)."""
        g, b = self._affine()
        if mean.shape[0] != x.shape[0]:       # tile-major rows sharing their image's statistics
            return ops.group_norm_apply_shared(x, mean, rstd, g, b, self.num_groups, act, split=split, also_cast=also_cast)
        return ops.group_norm_apply(x, mean, rstd, g, b, self.num_groups, act, split=split, also_cast=also_cast)

    def forward(self, x):  # NCHW (or [B, C, L]) compat
        shp = x.shape
        x4 = x.reshape(shp[0], shp[1], -1, 1) if x.dim() == 3 else x
        y = self.nhwc(ops.nchw_to_nhwc(x4.contiguous()))
        return ops.nhwc_to_nchw(y, dtype=ops.io_dtype(x)).to(x.dtype).reshape(shp)


class LayerNorm(nn.LayerNorm, _Packed):
    def _affine(self):
        if not self.elementwise_affine:
            return None, None
        return self._packed(lambda: (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous()),
                            self.weight, self.bias)

    def nhwc(self, x, a=None, b=None, split=1):
        if a is None and b is None:
            a, b = self._affine()
        return ops.layer_norm(x, a, b, self.eps, split=split)

    def forward(self, x):
        return self.nhwc(x.to(ops.stream_dtype()).contiguous()).to(x.dtype)


class RMSNormWeight(nn.Module):
    """Parameter holder for diffusers' RMSNorm (`weight` [dim]); applied inside omgsr_rmsnorm_rope."""

    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def w32(self):
        return self.weight.detach().float().contiguous()

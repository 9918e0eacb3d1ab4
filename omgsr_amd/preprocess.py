"""SURVEY §8(f) f2 — GPU counterpart of the reference driver's pre-process (infer/infer_omgsr_s.py:71-84,92 ==
infer/infer_omgsr_f.py:74-87,95): PIL `Image.resize` x `--upscale` (default resample of an RGB image: BICUBIC), then
`Image.resize(..., Image.LANCZOS)` to width / height rounded down to multiples of 8, then `to_tensor(img) * 2 - 1`.

    lq = preprocess_u8(image_u8, process_size=512, upscale=4)      # uint8 [B,H,W,3] on the device -> uint8 [B,H',W',3]
    x = colorfix.image_to_model_input(lq)                           # -> the model's NHWC input

Pillow's 8-bit resampler is reproduced BIT FOR BIT (tests/golden/pil_resize.npz holds Pillow's own outputs): the fixed-point tap
tables are computed here on the host in float64 with libm's sin — the same arithmetic as Pillow's C `precompute_coeffs` /
`normalize_coeffs_8bpc` — and the two passes (horizontal into an 8-bit temporary, then vertical; an unchanged axis is skipped)
run as HIP kernels (csrc/preprocess.hip, omgsr_resample_u8). Same names / argument meaning as the PIL calls they replace.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from functools import lru_cache
from typing import Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

BICUBIC, LANCZOS = "bicubic", "lanczos"
PRECISION_BITS = 32 - 8 - 2


def _bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc_filter(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos_filter(x: float) -> float:
    return _sinc_filter(x) * _sinc_filter(x / 3) if -3.0 <= x < 3.0 else 0.0


_FILTERS = {BICUBIC: (_bicubic_filter, 2.0), LANCZOS: (_lanczos_filter, 3.0)}


@lru_cache(maxsize=64)
def precompute_coeffs(in_size: int, out_size: int, filt: str) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size): (bounds [out,2], kk [out,ksize], ksize), int32."""
    if filt not in _FILTERS:
        raise ValueError(f"resample must be one of {sorted(_FILTERS)}; got {filt!r}")
    fn, base_support = _FILTERS[filt]
    scale = float(np.float32(in_size) - np.float32(0)) / out_size          # the box edges are C floats
    filterscale = scale if scale >= 1.0 else 1.0
    support = base_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        taps = [fn((x + lo - center + 0.5) * inv) for x in range(n)]
        total = 0.0
        for t in taps:
            total += t
        for x, t in enumerate(taps):
            if total != 0.0:
                t = t / total
            kk[xx, x] = int(-0.5 + t * one) if t < 0 else int(0.5 + t * one)
        bounds[xx, 0], bounds[xx, 1] = lo, n
    return bounds, kk, ksize


# Device copies of the tap tables: a bounded LRU (a folder of differently sized images - the reference driver's use case - asks for up
# to six new tables per image; an unbounded dict would pin all of them in HBM for the life of the process). The host-side loop above
# stays scalar on purpose: Pillow's taps go through libm's sin in float64 and a vectorised sin (numpy dispatches to SIMD kernels whose
# last ulp differs) would break the bit-exactness the goldens pin; its results are cached per (in, out, filter) too.
_DEV_TABLES_MAX = 64
_dev_tables: "OrderedDict[tuple, tuple]" = OrderedDict()


def _tables(in_size: int, out_size: int, filt: str, device):
    key = (in_size, out_size, filt, str(device))
    hit = _dev_tables.get(key)
    if hit is not None:
        _dev_tables.move_to_end(key)
        return hit
    b, k, ks = precompute_coeffs(in_size, out_size, filt)
    val = (torch.from_numpy(b).to(device).contiguous(), torch.from_numpy(k).to(device).contiguous(), ks)
    _dev_tables[key] = val
    while len(_dev_tables) > _DEV_TABLES_MAX:
        _dev_tables.popitem(last=False)           # the launches that used it are ordered before its free on the same stream
    return val


def _pass(img: torch.Tensor, out_size: int, axis: int, filt: str) -> torch.Tensor:
    N, H, W, _ = img.shape
    b, k, ks = _tables(W if axis == 1 else H, out_size, filt, img.device)
    out = torch.empty((N, out_size if axis == 0 else H, out_size if axis == 1 else W, 3), device=img.device, dtype=torch.uint8)
    check(_lib.load().omgsr_resample_u8(img.data_ptr(), out.data_ptr(), b.data_ptr(), k.data_ptr(), ks, N, H, W, out_size, axis,
                                        ops._stream()), "omgsr_resample_u8")
    return out


def resize_u8(image_u8: torch.Tensor, size: Tuple[int, int], resample: str = BICUBIC) -> torch.Tensor:
    """`Image.resize(size, resample)` on a batch of RGB images: uint8 [B,H,W,3] on the device, size = (width, height)."""
    ops._req(image_u8, torch.uint8, "image_u8")
    if image_u8.dim() != 4 or image_u8.shape[-1] != 3:
        raise ValueError("image_u8 must be [B,H,W,3]")
    ow, oh = int(size[0]), int(size[1])
    if ow <= 0 or oh <= 0:
        raise ValueError("height and width must be > 0")
    out = image_u8
    if ow != image_u8.shape[2]:
        out = _pass(out, ow, 1, resample)
    if oh != image_u8.shape[1]:
        out = _pass(out, oh, 0, resample)
    return out.clone() if out is image_u8 else out


def preprocess_u8(image_u8: torch.Tensor, process_size: int = 512, upscale: int = 4) -> torch.Tensor:
    """The resize chain of infer/infer_omgsr_s.py:71-84 on uint8 [B,H,W,3]: inputs smaller than process_size // upscale are first
    brought up to it (bicubic), then x upscale (bicubic), then snapped to multiples of 8 (LANCZOS)."""
    _, h, w, _ = image_u8.shape
    if w < process_size // upscale or h < process_size // upscale:
        scale = (process_size // upscale) / min(w, h)
        image_u8 = resize_u8(image_u8, (int(scale * w), int(scale * h)), BICUBIC)
        _, h, w, _ = image_u8.shape
    image_u8 = resize_u8(image_u8, (w * upscale, h * upscale), BICUBIC)
    _, h, w, _ = image_u8.shape
    return resize_u8(image_u8, (w - w % 8, h - h % 8), LANCZOS)

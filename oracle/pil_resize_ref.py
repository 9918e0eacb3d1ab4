"""ORACLE — test infrastructure only (never imported by omgsr_amd/).

numpy restatement of Pillow's `Image.resize` for 8-bit RGB images (`ImagingResample`, libImaging/Resample.c of the Pillow the
reference runs on: `requirements.txt` pins no version, the image here has Pillow 12.2) with the two filters the reference's
driver uses: BICUBIC (PIL's default `resample` for RGB, infer/infer_omgsr_s.py:78,81) and LANCZOS (the /8 snap, :84).

Algorithm (published in Pillow's source; restated):
  * per axis, for every output index xx: center = (xx + 0.5) * scale, scale = in / out, filterscale = max(scale, 1),
    support = filter.support * filterscale; taps xmin = trunc(center - support + 0.5) clipped at 0 .. xmax = trunc(center +
    support + 0.5) clipped at in; weight k[x] = filter((x + xmin - center + 0.5) / filterscale), normalised by their sum
  * 8-bit images: weights become fixed point, kk = trunc(k * 2^22 +- 0.5) (PRECISION_BITS = 32 - 8 - 2); a pass accumulates
    2^21 + sum(pixel * kk) in int32, shifts right by 22 and clips to 0..255
  * two passes, HORIZONTAL first (into an 8-bit temporary), then vertical; a pass whose size does not change is skipped

PINNED: tests/golden/pil_resize.npz holds outputs of PIL itself (tests/golden/make_golden_pil_resize.py, run in the build
container where Pillow is installed); tests/test_pil_resize_golden.py checks this file against them bit for bit.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BICUBIC, LANCZOS = "bicubic", "lanczos"
SUPPORT = {BICUBIC: 2.0, LANCZOS: 3.0}


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


FILTER = {BICUBIC: _bicubic, LANCZOS: _lanczos}


def precompute_coeffs(in_size: int, out_size: int, filt: str):
    """-> (bounds [out, 2] int32 (xmin, count), kk [out, ksize] int32 fixed-point weights, ksize)."""
    f, support0 = FILTER[filt], SUPPORT[filt]
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size        # (double)(in1 - in0) / outSize with float box edges
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img: np.ndarray, out_size: int, filt: str, axis: int) -> np.ndarray:
    """img uint8 [H, W, C]; resample along `axis` (0 vertical, 1 horizontal)."""
    in_size = img.shape[axis]
    bounds, kk, ksize = precompute_coeffs(in_size, out_size, filt)
    src = np.moveaxis(img, axis, 0).astype(np.int64)                       # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[xmin:xmin + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize(img: np.ndarray, size, filt: str = BICUBIC) -> np.ndarray:
    """`Image.fromarray(img).resize(size, resample)` for uint8 RGB: size = (width, height) like PIL."""
    ow, oh = size
    H, W, _ = img.shape
    out = img
    if ow != W:
        out = _pass(out, ow, filt, 1)
    if oh != H:
        out = _pass(out, oh, filt, 0)
    return out.copy() if out is img else out


def driver_preprocess(img: np.ndarray, process_size: int = 512, upscale: int = 4) -> np.ndarray:
    """infer/infer_omgsr_s.py:71-84 on an RGB uint8 array: optional up-resize of small inputs, x`upscale` bicubic, LANCZOS snap
    of width / height down to multiples of 8."""
    h, w, _ = img.shape
    if w < process_size // upscale or h < process_size // upscale:
        scale = (process_size // upscale) / min(w, h)
        img = resize(img, (int(scale * w), int(scale * h)), BICUBIC)
        h, w, _ = img.shape
    img = resize(img, (w * upscale, h * upscale), BICUBIC)
    h, w, _ = img.shape
    return resize(img, (w - w % 8, h - h % 8), LANCZOS)

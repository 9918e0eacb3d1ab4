"""Latent tiling shared by OMGSR-S and OMGSR-F: tile grid, Gaussian blending weights and the fp32
weighted stitch of per-tile denoiser outputs.

Behavioural counterpart of infer/omgsr_s_infer_model.py:56-71,88-161 (== infer/omgsr_f_infer_model.py:
157-172,214-314), including the quirks recorded in SURVEY.md Appendix C-2/C-3:
  * `grid_rows` is derived from the WIDTH and drives ofs_x, `grid_cols` from the HEIGHT and drives ofs_y
  * the last row/col tile is flushed to the far edge
  * Gaussian weights: var 0.01, x-midpoint (w-1)/2 but y-midpoint h/2 (vertically asymmetric), built in fp64
  * accumulation in fp32 buffers, normalised by the accumulated weights
"""
from __future__ import annotations

from math import exp, pi, sqrt
from typing import Callable, List, Tuple

import numpy as np
import torch

from .. import ops


def gaussian_weights(tile_width: int, tile_height: int) -> np.ndarray:
    """[tile_height, tile_width] fp64 (reference: _gaussian_weights, before the torch.tile over batch/channels)."""
    var = 0.01
    mid = (tile_width - 1) / 2
    x_probs = [exp(-(x - mid) * (x - mid) / (tile_width * tile_width) / (2 * var)) / sqrt(2 * pi * var) for x in range(tile_width)]
    mid = tile_height / 2
    y_probs = [exp(-(y - mid) * (y - mid) / (tile_height * tile_height) / (2 * var)) / sqrt(2 * pi * var) for y in range(tile_height)]
    return np.outer(y_probs, x_probs)


def tile_grid(h: int, w: int, tile_size: int, tile_overlap: int) -> Tuple[int, List[Tuple[int, int]]]:
    """Returns (effective tile size, [(ofs_y, ofs_x), ...]) in the reference's visiting order
    (outer loop `row` over the width-derived count, inner loop `col` over the height-derived count)."""
    tile_size = min(tile_size, min(h, w))
    grid_rows, cur = 0, 0
    while cur < w:
        cur = max(grid_rows * tile_size - tile_overlap * grid_rows, 0) + tile_size
        grid_rows += 1
    grid_cols, cur = 0, 0
    while cur < h:
        cur = max(grid_cols * tile_size - tile_overlap * grid_cols, 0) + tile_size
        grid_cols += 1
    out = []
    for row in range(grid_rows):
        for col in range(grid_cols):
            ofs_x = w - tile_size if row == grid_rows - 1 else max(row * tile_size - tile_overlap * row, 0)
            ofs_y = h - tile_size if col == grid_cols - 1 else max(col * tile_size - tile_overlap * col, 0)
            out.append((ofs_y, ofs_x))
    return tile_size, out


_WTS: dict = {}


def _weights_on(ts: int, device) -> torch.Tensor:
    """The Gaussian tile weights as a device tensor, uploaded once per (tile size, device): a host -> device copy per call would also be
    illegal inside a hipGraph capture (pipelines/graphed.py)."""
    key = (ts, str(device))
    t = _WTS.get(key)
    if t is None:
        t = _WTS[key] = torch.tensor(gaussian_weights(ts, ts), dtype=torch.float32, device=device)
    return t


def tiled_denoise(latent_nhwc: torch.Tensor, channels: int, tile_size: int, tile_overlap: int,
                  denoise: Callable[[torch.Tensor], torch.Tensor], tiles_per_call: int = 16) -> torch.Tensor:
    """latent_nhwc [B,h,w,C8] stream tensor; `denoise(tiles [n*B,t,t,C8]) -> [n*B,t,t,>=channels]`. Returns the
    Gaussian-blended prediction [B,h,w,C8] of the same element kind (channels >= `channels` zero).

    The reference runs one denoiser call per tile (its "batching" never batches, SURVEY C-4). Tiles are
    independent samples, so here up to `tiles_per_call` tiles ride in ONE call along the batch axis — the
    result per tile is bit-identical (tests/test_models_gpu.py::test_unet_forward checks batch == B x batch-1)
    while the low-resolution UNet layers see 9x more rows and fill the 256 CUs. Stitch order is unchanged."""
    B, h, w, ld = latent_nhwc.shape
    ts, offsets = tile_grid(h, w, tile_size, tile_overlap)
    wts = _weights_on(ts, latent_nhwc.device)
    acc = torch.zeros((B, h, w, channels), device=latent_nhwc.device, dtype=torch.float32)
    wsum = torch.zeros((1, h, w, 1), device=latent_nhwc.device, dtype=torch.float32)
    for i0 in range(0, len(offsets), max(1, tiles_per_call)):
        group = offsets[i0:i0 + max(1, tiles_per_call)]
        tiles = [ops.crop_nhwc(latent_nhwc, oy, ox, ts, ts) for (oy, ox) in group]
        preds = denoise(torch.cat(tiles, 0) if len(tiles) > 1 else tiles[0])
        for j, (oy, ox) in enumerate(group):
            ops.tile_accumulate(preds[j * B:(j + 1) * B].contiguous(), wts, acc, oy, ox)
            ops.tile_accumulate(None, wts, wsum, oy, ox)
    return ops.tile_normalise(acc, wsum, ld=ld, dtype=latent_nhwc.dtype)

"""ORACLE — test infrastructure only (never imported by omgsr_amd/).

CPU fp32 restatement of the reference's tiled VAE (infer/vaehook.py:371-427,459-634,681-829, "exact"
mode and the fast mode of :637-677,714-735) written layer-synchronously instead of as a task queue:

  * split the input into tiles (input bbox expanded by pad = 32 encoder / 11 decoder, clipped) and the
    matching output bboxes (:577-634)
  * run every tile through the same op list; at EVERY GroupNorm stop, collect each tile's per-(n, group)
    biased (var, mean) (:371-381), merge them with weights proportional to the tile's pixel count - a
    weighted mean of variances and of means, NOT the pooled variance (:489-508) - and normalise every tile
    with the merged statistics, eps 1e-6 (:384-413)
  * the mid-block attention is computed inside each tile (:137-171,245)
  * crop each tile's valid region and paste it, no blending (:416-427,803-805); the result buffer is fp32

Pinned by tests/golden/vaehook.npz, produced by running the REFERENCE's VAEHook (imported from
/root/reference) on the oracle's reduced Encoder / Decoder (tests/golden/make_golden_vaehook.py).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn.functional as F


def get_best_tile_size(lowerbound: int, upperbound: int) -> int:       # :562-575
    divider = 32
    while divider >= 2:
        rem = lowerbound % divider
        if rem == 0:
            return lowerbound
        cand = lowerbound - rem + divider
        if cand <= upperbound:
            return cand
        divider //= 2
    return lowerbound


def split_tiles(h: int, w: int, tile_size: int, pad: int, is_decoder: bool) -> Tuple[List[List[int]], List[List[int]]]:
    """bboxes are [x1, x2, y1, y2]."""                                   # :577-634
    nh = max(math.ceil((h - 2 * pad) / tile_size), 1)
    nw = max(math.ceil((w - 2 * pad) / tile_size), 1)
    th = get_best_tile_size(math.ceil((h - 2 * pad) / nh), tile_size)
    tw = get_best_tile_size(math.ceil((w - 2 * pad) / nw), tile_size)
    ins, outs = [], []
    for i in range(nh):
        for j in range(nw):
            ib = [pad + j * tw, min(pad + (j + 1) * tw, w), pad + i * th, min(pad + (i + 1) * th, h)]
            ob = [ib[0] if ib[0] > pad else 0, ib[1] if ib[1] < w - pad else w,
                  ib[2] if ib[2] > pad else 0, ib[3] if ib[3] < h - pad else h]
            outs.append([x * 8 if is_decoder else x // 8 for x in ob])
            ins.append([max(0, ib[0] - pad), min(w, ib[1] + pad), max(0, ib[2] - pad), min(h, ib[3] + pad)])
    return ins, outs


def group_var_mean(x: torch.Tensor, groups: int = 32):
    """per (n, group) biased variance and mean -> [n*groups] each."""     # :371-381
    b, c = x.shape[:2]
    xr = x.reshape(b * groups, -1)
    return xr.var(dim=1, unbiased=False), xr.mean(dim=1)


def fixed_group_norm(x, mean, var, weight, bias, groups: int = 32, eps: float = 1e-6):   # :384-413
    b, c = x.shape[:2]
    xr = x.reshape(b * groups, -1)
    y = ((xr - mean[:, None]) / torch.sqrt(var[:, None] + eps)).reshape(x.shape)
    return y * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def merge_stats(vars_: List[torch.Tensor], means: List[torch.Tensor], pixels: List[int]):    # :489-508
    p = torch.tensor(pixels, dtype=torch.float32) / max(pixels)
    p = (p / p.sum())[:, None]
    return (torch.vstack(vars_) * p).sum(0), (torch.vstack(means) * p).sum(0)


def _attention_in_tile(attn, h):                                          # :137-171
    b, c, hh, ww = h.shape
    t = h.view(b, c, hh * ww).transpose(1, 2)
    q, k, v = attn.to_q(t), attn.to_k(t), attn.to_v(t)
    p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * attn.scale, dim=-1)
    o = attn.to_out[0](torch.bmm(p, v))
    return o.transpose(-1, -2).reshape(b, c, hh, ww)


def op_list(net, is_decoder: bool):
    """Flat op list, same order as build_task_queue (:332-359). Entries: ('gn', norm, silu?) are the
    cross-tile barriers; ('f', fn) are per-tile ops; ('res_push', fn) / ('res_add',) bracket residual paths."""
    ops = [("f", net.conv_in)]

    def resblock(b):
        ops.append(("res_push", b.conv_shortcut if b.in_channels != b.out_channels else (lambda x: x)))
        ops.extend([("gn", b.norm1, True), ("f", b.conv1), ("gn", b.norm2, True), ("f", b.conv2), ("res_add",)])

    def attn(a):
        ops.append(("res_push", lambda x: x))
        ops.extend([("gn", a.group_norm, False), ("f", lambda x, a=a: _attention_in_tile(a, x)), ("res_add",)])

    def mid():
        resblock(net.mid_block.resnets[0]); attn(net.mid_block.attentions[0]); resblock(net.mid_block.resnets[1])

    if is_decoder:
        mid()
        blocks = net.up_blocks
    else:
        blocks = net.down_blocks
    for i, blk in enumerate(blocks):
        for r in blk.resnets:
            resblock(r)
        if i != len(blocks) - 1:
            ops.append(("f", blk.upsamplers[0] if is_decoder else blk.downsamplers[0]))
    if not is_decoder:
        mid()
    ops.extend([("gn", net.conv_norm_out, True), ("f", net.conv_out)])
    return ops


@torch.no_grad()
def tiled_forward(net, x: torch.Tensor, tile_size: int, is_decoder: bool, fast: bool = False) -> torch.Tensor:
    """VAEHook.__call__ (:548-560) + vae_tile_forward (:681-829)."""
    pad = 11 if is_decoder else 32
    N, _, H, W = x.shape
    if max(H, W) <= pad * 2 + tile_size:
        return net(x)
    ins, outs = split_tiles(H, W, tile_size, pad, is_decoder)
    tiles = [x[:, :, b[2]:b[3], b[0]:b[1]].clone() for b in ins]
    ops = op_list(net, is_decoder)
    fixed = None
    if fast:                                                             # :714-735 + estimate_group_norm :637-677
        scale = tile_size / max(H, W)
        small = F.interpolate(x, scale_factor=scale, mode="nearest-exact")
        std_o, mean_o = torch.std_mean(x, dim=[0, 2, 3], keepdim=True)
        std_n, mean_n = torch.std_mean(small, dim=[0, 2, 3], keepdim=True)
        small = ((small - mean_n) / std_n * std_o + mean_o).clamp_(min=x.min(), max=x.max())
        fixed, t, res = [], small, []
        for op in ops:
            if op[0] == "gn":
                v, m = group_var_mean(t)
                fixed.append((v, m))
                t = fixed_group_norm(t, m, v, op[1].weight, op[1].bias)
                if op[2]:
                    t = F.silu(t)
            elif op[0] == "f":
                t = op[1](t)
            elif op[0] == "res_push":
                res.append(op[1](t))
            else:
                t = t + res.pop()
    res_stack = [[] for _ in tiles]
    gi = 0
    for op in ops:
        if op[0] == "gn":
            if fixed is not None:
                var, mean = fixed[gi]
            else:
                stats = [group_var_mean(t) for t in tiles]
                var, mean = merge_stats([s[0] for s in stats], [s[1] for s in stats], [t.shape[2] * t.shape[3] for t in tiles])
            gi += 1
            tiles = [fixed_group_norm(t, mean, var, op[1].weight, op[1].bias) for t in tiles]
            if op[2]:
                tiles = [F.silu(t) for t in tiles]
        elif op[0] == "f":
            tiles = [op[1](t) for t in tiles]
        elif op[0] == "res_push":
            for i, t in enumerate(tiles):
                res_stack[i].append(op[1](t))
        else:
            tiles = [t + res_stack[i].pop() for i, t in enumerate(tiles)]
    Ho, Wo = (H * 8, W * 8) if is_decoder else (H // 8, W // 8)
    result = torch.zeros((N, tiles[0].shape[1], Ho, Wo), dtype=torch.float32)
    for t, ib, ob in zip(tiles, ins, outs):                              # crop_valid_region :416-427
        pb = [v * 8 if is_decoder else v // 8 for v in ib]
        mg = [ob[i] - pb[i] for i in range(4)]
        result[:, :, ob[2]:ob[3], ob[0]:ob[1]] = t[:, :, mg[2]:t.shape[2] + mg[3], mg[0]:t.shape[3] + mg[1]]
    return result

"""Tiled VAE on MI355X — counterpart of infer/vaehook.py::VAEHook (same constructor arguments, same
`__call__` dispatch, same tile geometry and cross-tile GroupNorm semantics), re-designed for one GPU
with 288 GB of HBM:

  * the reference parks every tile but one on the CPU between its 22 (encoder) / 30 (decoder) GroupNorm
    barriers and walks a Python task queue per tile (PCIe-bound); here ALL tiles stay resident in HBM, tiles
    of equal shape are stacked along the batch axis (one dense tensor per shape group: corner / edge / interior
    tiles), and every conv layer runs as ONE launch over all groups (ops.conv2d_multi -> omgsr_igemm_multi: a
    prefix table of the groups' tile counts in the kernel arguments) - no group pays its own partial last round
    of workgroups, and small groups still take the halo-tile kernel
  * at every GroupNorm the per-tile (mean, biased var) come from the statistics kernel, are merged per image
    with pixel-count weights exactly like GroupNormParam.summary() (a weighted mean of variances, NOT the
    pooled variance), and the apply(+SiLU) kernel runs with the externally supplied statistics (eps 1e-6)
  * mid-block attention stays tile-local, valid regions are cropped and pasted without blending
Fast mode (fast_encoder / fast_decoder) estimates every GroupNorm's statistics once on a nearest-exact
down-sampled, re-standardised copy of the whole input, then all tiles use those fixed statistics.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .. import ops


def get_best_tile_size(lowerbound: int, upperbound: int) -> int:
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
    """(input bboxes, output bboxes), each [x1, x2, y1, y2]; behaviour of VAEHook.split_tiles (infer/vaehook.py:577-634)."""
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
            outs.append([v * 8 if is_decoder else v // 8 for v in ob])
            ins.append([max(0, ib[0] - pad), min(w, ib[1] + pad), max(0, ib[2] - pad), min(h, ib[3] + pad)])
    return ins, outs


class VAEHook:
    def __init__(self, net, tile_size, is_decoder, fast_decoder, fast_encoder, color_fix, to_gpu=False):
        self.net = net                      # omgsr_amd Encoder | Decoder
        self.tile_size = tile_size
        self.is_decoder = is_decoder
        self.fast_mode = (fast_encoder and not is_decoder) or (fast_decoder and is_decoder)
        self.color_fix = color_fix and not is_decoder
        self.to_gpu = to_gpu
        self.pad = 11 if is_decoder else 32

    # ---- op list (order of build_task_queue, infer/vaehook.py:332-359) ----------------------
    def _ops(self):
        net, dec = self.net, self.is_decoder
        # ("conv", module, kwargs): module.nhwc_multi over every shape group in one launch
        seq = [("conv", net.conv_in, dict(gn_groups=net.conv_norm_out.num_groups))]

        # ("gn", norm, act, consumer): the normalised tensor is the consumer's MFMA operand (its in_split() picks the plain /
        # two-term split form in the accurate tier)
        # ("gn", ..., shortcut): the block's 1x1 conv_shortcut reads the same tensor norm1 normalises; its operand copy is a
        # second output of the apply pass and the shortcut result is pushed as the residual (same values as infer/vaehook.py's
        # ('store_res', conv_shortcut) -> ('pre_norm', norm1) order: neither op changes x)
        # ("conv_res", conv2, gn_groups, out_for): out_for = the up / down-sampling conv that is the result's only consumer
        def resblock(b, out_for=None):
            if b.conv_shortcut is None:
                seq.append(("res_push", None))
            seq.append(("gn", b.norm1, ops.ACT_SILU, b.conv1, b.conv_shortcut))
            seq.append(("conv", b.conv1, dict(gn_groups=b.norm2.num_groups)))
            seq.append(("gn", b.norm2, ops.ACT_SILU, b.conv2, None))
            seq.append(("conv_res", b.conv2, b.norm1.num_groups, out_for))  # conv2 + residual in the GEMM epilogue (+ next GN statistics)

        def attn(a):
            seq.append(("res_push", None))
            seq.append(("gn", a.group_norm, ops.ACT_NONE, a.to_q, None))
            seq.append(("attn_res", a))

        def mid():
            resblock(net.mid_block.resnets[0]); attn(net.mid_block.attentions[0]); resblock(net.mid_block.resnets[1])

        if dec:
            mid()
            blocks = net.up_blocks
        else:
            blocks = net.down_blocks
        for i, blk in enumerate(blocks):
            samp = None if i == len(blocks) - 1 else (blk.upsamplers[0] if dec else blk.downsamplers[0])
            for j, r in enumerate(blk.resnets):
                resblock(r, out_for=samp.conv if (samp is not None and j == len(blk.resnets) - 1) else None)
            if samp is not None:
                g = blk.resnets[0].norm1.num_groups
                if dec:     # Upsample2D: nearest-2x folded into the conv; Downsample2D (VAE): F.pad(x, (0, 1, 0, 1)) + a valid stride-2 conv
                    seq.append(("conv", samp.conv, dict(upsample=True, gn_groups=g)))
                else:
                    seq.append(("conv", samp.conv, dict(pad=(0, 1, 0, 1) if samp.padding == 0 else samp.padding, gn_groups=g)))
        if not dec:
            mid()
        seq.append(("gn", net.conv_norm_out, ops.ACT_SILU, net.conv_out, None))
        seq.append(("conv", net.conv_out, {}))
        return seq

    # ---- public entry (NHWC bf16) -------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        N, H, W, _ = x.shape
        if max(H, W) <= self.pad * 2 + self.tile_size:
            return self.net.nhwc(x)            # "[Tiled VAE]: the input size is tiny and unnecessary to tile."
        return self.vae_tile_forward(x)

    def _run(self, seq, groups: Dict[tuple, torch.Tensor], counts: Dict[tuple, int], N: int, fixed=None, record=None):
        """Layer-synchronous execution over shape groups. groups[shape] = [T_g*N, h, w, C] (tile-major)."""
        res: Dict[tuple, list] = {k: [] for k in groups}
        gi = 0
        pending = None          # a GroupNorm handed to the conv that follows it (ops.GnSpec): the conv's patch producer where the kernel can
        for op in seq:
            kind = op[0]
            if kind == "gn":
                norm, act = op[1], op[2]
                G = norm.num_groups
                if fixed is not None:
                    mean, var = fixed[gi]
                    rstd = torch.rsqrt(var + 1e-6)                 # custom_group_norm: eps 1e-6
                else:
                    # per-tile statistics merged per image with pixel-count weights (GroupNormParam.summary), one launch:
                    # the partials come from the producing conv's epilogue where it left them
                    keys = list(groups)
                    mean, rstd, var = ops.group_norm_stats_merged([groups[k] for k in keys], [counts[k] for k in keys], N, G, 1e-6)
                if record is not None:
                    record.append((mean, var))
                gi += 1
                first = next(iter(groups.values()))
                is_conv3 = getattr(op[3], "kernel_size", None) == (3, 3)
                if is_conv3 and first.dtype != torch.float32:
                    # a 16-bit stream tensor into a 3x3 conv: the conv takes the norm (fused into its patch producer, or applied in front of
                    # it: ops.conv2d_multi decides for the whole layer); the 1x1 shortcut reads the un-normalised tensor, which IS its operand
                    pending = norm.spec_stats(mean, rstd, act)
                    if op[4] is not None:
                        for k in groups:
                            res[k].append(op[4].nhwc(groups[k], pad=0))
                    continue
                for k in groups:       # rows (tile, image) share the image's statistics
                    if op[4] is not None:
                        groups[k], xc = norm.apply_stats(groups[k], mean, rstd, act, split=op[3].in_split(), also_cast=op[4].in_split())
                        res[k].append(op[4].nhwc(xc, pad=0))
                    else:
                        groups[k] = norm.apply_stats(groups[k], mean, rstd, act, split=op[3].in_split())
            elif kind == "conv":
                keys = list(groups)
                for k, y in zip(keys, op[1].nhwc_multi([groups[k] for k in keys], gn=pending, **op[2])):
                    groups[k] = y
                pending = None
            elif kind == "res_push":
                for k in groups:
                    res[k].append(op[1](groups[k]) if op[1] is not None else groups[k])
            elif kind == "conv_res":
                keys = list(groups)
                xs, rs = [groups[k] for k in keys], [res[k].pop() for k in keys]
                if op[3] is not None:
                    outs = op[1].nhwc_multi(xs, residuals=rs, out_dtype=ops.OUT_BF16, out_split=op[3].in_split(), gn=pending)
                else:
                    outs = op[1].nhwc_multi(xs, residuals=rs, gn_groups=op[2], gn=pending)
                pending = None
                for k, y in zip(keys, outs):
                    groups[k] = y
            elif kind == "attn_res":
                for k in groups:
                    groups[k] = op[1].attend(groups[k], residual=res[k].pop())
        return groups

    @torch.no_grad()
    def vae_tile_forward(self, x: torch.Tensor) -> torch.Tensor:
        N, H, W, _ = x.shape
        ins, outs = split_tiles(H, W, self.tile_size, self.pad, self.is_decoder)
        seq = self._ops()
        fixed = None
        if self.fast_mode:
            fixed = self._estimate(x, seq)
        # stack tiles of equal shape along the batch axis (tile-major)
        order: Dict[tuple, List[int]] = {}
        for i, b in enumerate(ins):
            order.setdefault((b[3] - b[2], b[1] - b[0]), []).append(i)
        groups = {k: torch.cat([ops.crop_nhwc(x, ins[i][2], ins[i][0], k[0], k[1]) for i in idx], 0) for k, idx in order.items()}
        counts = {k: len(idx) for k, idx in order.items()}
        groups = self._run(seq, groups, counts, N, fixed=fixed)
        Ho, Wo = (H * 8, W * 8) if self.is_decoder else (H // 8, W // 8)
        Cout = next(iter(groups.values())).shape[-1]
        result = torch.zeros((N, Ho, Wo, Cout), device=x.device, dtype=next(iter(groups.values())).dtype)
        for k, idx in order.items():
            t = groups[k]
            for j, i in enumerate(idx):
                ib, ob = ins[i], outs[i]
                pb = [v * 8 if self.is_decoder else v // 8 for v in ib]
                mg = [ob[q] - pb[q] for q in range(4)]                 # crop_valid_region
                th, tw = ob[3] - ob[2], ob[1] - ob[0]
                ops.paste_nhwc(t[j * N:(j + 1) * N], result, mg[2], mg[0], ob[2], ob[0], th, tw)
        return result

    def _estimate(self, x: torch.Tensor, seq):
        """Fast mode (infer/vaehook.py:714-735, 637-677): statistics of every GroupNorm from one pass over a
        nearest-exact down-sampled copy of the input whose per-channel mean/std are restored. The resampling and
        re-standardisation act on one small tensor and use torch (plumbing); the network pass uses the HIP kernels."""
        N, H, W, Cp = x.shape
        scale = self.tile_size / max(H, W)
        # the down-sampled copy comes from the library's own gather kernel (F.interpolate's nearest-exact index formula, ops.resize_nearest_exact);
        # the per-channel moments of the two tensors are reductions over one small tensor (torch, plumbing - like the reference's own)
        xc = x.float()
        small = ops.resize_nearest_exact(x, scale).float()
        std_o, mean_o = torch.std_mean(xc, dim=[0, 1, 2], keepdim=True)
        std_n, mean_n = torch.std_mean(small, dim=[0, 1, 2], keepdim=True)
        std_n = torch.where(std_n == 0, torch.ones_like(std_n), std_n)      # zero-padded channels
        small = ((small - mean_n) / std_n * std_o + mean_o).clamp_(min=float(xc.min()), max=float(xc.max()))
        small = small.to(ops.stream_dtype()).contiguous()
        record: list = []
        self._run(seq, {(small.shape[1], small.shape[2]): small}, {(small.shape[1], small.shape[2]): 1}, N, record=record)
        return record

"""AutoencoderKL with diffusers' API surface (config fields, attribute tree, state-dict keys,
encode()/decode() signatures) running on the gfx950 kernels.

Stands in for `diffusers.AutoencoderKL` at the reference call sites
infer/omgsr_s_infer_model.py:11,84,166,173 and infer/omgsr_f_infer_model.py:16,99,211,318; the
attribute tree is the one infer/vaehook.py:296-329,340-355 duck-types.

Data layout: NCHW tensors at the API boundary (as diffusers), bf16 NHWC inside. GroupNorm+SiLU runs
as a statistics pass + an apply pass feeding the implicit-GEMM conv; nearest-2x upsampling and the
encoder's asymmetric pad are folded into the conv's gather; the d=512 single-head mid attention
uses the batched-GEMM + masked row-softmax path.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..nn import Conv2d, GroupNorm, Linear
from .modeling_utils import ConfigDict, ModelMixin

SD21_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                       latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, shift_factor=None,
                       use_quant_conv=True, use_post_quant_conv=True, act_fn="silu", sample_size=768)
FLUX_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                       latent_channels=16, norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159,
                       use_quant_conv=False, use_post_quant_conv=False, act_fn="silu", sample_size=1024)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: Optional[int], groups: int, eps: float):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = Conv2d(out_channels, out_channels, 3, padding=1)
        self.use_in_shortcut = in_channels != out_channels
        self.conv_shortcut = Conv2d(in_channels, out_channels, 1) if self.use_in_shortcut else None
        self.nonlinearity = nn.SiLU()

    def nhwc(self, x, conv1_bias=None, out_for=None):
        """conv1_bias: conv1.bias + time_emb_proj(silu(temb)) folded by the UNet (constant at fixed t*).
        out_for: the ONLY consumer of the result is that conv / linear (an up / down-sampling conv): the result is written
        directly as its MFMA operand (no fp32 stream copy, no cast pass)."""
        # norm -> SiLU -> conv: the norm is handed to the conv (gn=): it runs as the conv's patch producer where the kernel can (a 16-bit
        # stream tensor into a plain 3x3 conv on the halo-tile kernel: every resnet conv of the fast tiers), as the apply pass otherwise
        if self.conv_shortcut is not None and x.dtype == torch.float32:
            # accurate tier: the 1x1 shortcut reads x itself; its operand copy is a second output of norm1's apply pass over x
            h, xc = self.norm1.nhwc(x, ops.ACT_SILU, split=self.conv1.in_split(), also_cast=self.conv_shortcut.in_split())
            h = self.conv1.nhwc(h, bias_override=conv1_bias, gn_groups=self.norm2.num_groups)
        else:                                                      # (a 16-bit stream tensor IS the shortcut's operand)
            xc = x
            h = self.conv1.nhwc(x, gn=self.norm1.spec(x, ops.ACT_SILU), bias_override=conv1_bias, gn_groups=self.norm2.num_groups)   # norm2's statistics ride the epilogue
        g2 = self.norm2.spec(h, ops.ACT_SILU)
        sc = self.conv_shortcut.nhwc(xc, pad=0) if self.conv_shortcut is not None else x
        if out_for is not None:
            return self.conv2.nhwc(h, gn=g2, residual=sc, out_dtype=ops.OUT_BF16, out_split=out_for.in_split())
        return self.conv2.nhwc(h, gn=g2, residual=sc, gn_groups=self.norm1.num_groups)      # ... and the next block's norm1

    def forward(self, x, temb=None):  # NCHW (diffusers calling convention; the VAE variant has no temb)
        if temb is not None or self.time_emb_proj is not None:
            raise NotImplementedError("ResnetBlock2D.forward: the UNet folds the time embedding at fixed t* (UNet2DConditionModel.nhwc)")
        return _nchw_call(self.nhwc, x, self.out_channels)


class Downsample2D(nn.Module):
    def __init__(self, channels: int, padding: int):
        super().__init__()
        self.padding = padding
        self.conv = Conv2d(channels, channels, 3, stride=2, padding=padding)

    def nhwc(self, x, gn_groups: int = 0):
        # VAE: F.pad(x, (0,1,0,1)) then a valid stride-2 conv; UNet: symmetric padding 1
        pad = (0, 1, 0, 1) if self.padding == 0 else self.padding
        return self.conv.nhwc(x, pad=pad, gn_groups=gn_groups)

    def forward(self, x):  # NCHW: what infer/vaehook.py:318-323 calls as `module[i_level].downsamplers[0]`
        return _nchw_call(self.nhwc, x, self.conv.out_channels)


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = Conv2d(channels, channels, 3, padding=1)
        self.conv.phase_upsample = True

    def nhwc(self, x, gn_groups: int = 0):
        """gn_groups: groups of the GroupNorm that consumes the result directly (VAE decoder: the next block's norm1)."""
        return self.conv.nhwc(x, upsample=True, gn_groups=gn_groups)

    def forward(self, x, output_size=None):  # NCHW (`module[i_level].upsamplers[0]`)
        if output_size is not None:
            raise NotImplementedError("Upsample2D.forward: explicit output_size (odd latent shapes) is outside the hot path (SURVEY A.6)")
        return _nchw_call(self.nhwc, x, self.conv.out_channels)


class VaeAttention(nn.Module):
    """diffusers Attention as built by the VAE mid block: 1 head of 512, GroupNorm, biased projections,
    residual connection."""

    def __init__(self, channels: int, groups: int):
        super().__init__()
        self.heads = 1
        self.scale = channels ** -0.5
        self.group_norm = GroupNorm(groups, channels, eps=1e-6)
        self.to_q = Linear(channels, channels)
        self.to_k = Linear(channels, channels)
        self.to_v = Linear(channels, channels)
        self.to_out = nn.ModuleList([Linear(channels, channels), nn.Dropout(0.0)])
        self.norm_cross = None

    def qkv_split(self) -> int:
        """Split of the operand the three projections share (the GroupNorm output)."""
        return self.to_q.in_split()

    def nhwc(self, x):
        return self.attend(self.group_norm.nhwc(x, split=self.qkv_split()), x)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):  # NCHW, with the residual
        if encoder_hidden_states is not None or attention_mask is not None:
            raise NotImplementedError("VaeAttention is the VAE mid block's unmasked self-attention")
        return _nchw_call(self.nhwc, hidden_states, hidden_states.shape[1])

    def attend(self, g, residual):
        """g: GroupNorm'ed operand [N,H,W,C*split] (the tiled VAE supplies cross-tile statistics); returns residual + attn."""
        N, H, W, Cg = g.shape
        Cc = residual.shape[-1]
        L = H * W
        x = residual
        g = g.reshape(N, L, Cg)
        Lp = ops._round_up(L, 128)
        # qk_split (accurate tier, omgsr_amd/precision.py VAE_QK_SPLIT): q and k leave their projections as two-term splits and the
        # score GEMM runs q_hi k_hi + q_lo k_hi + q_hi k_lo - the logits of a 4096-key softmax over ONE 512-wide head are where this
        # block's rounding is amplified (13 of the 44 units of the worst of 80 draws; tests/emulate_numerics.py --attn-exact dec:qk)
        # Range-fallback tier (ops.attn_split: bf16 operands, 8-bit mantissas): q and k split in BOTH attentions, and the probabilities and V as
        # two-term splits too, so the PV product runs p_hi v_hi + p_lo v_hi + p_hi v_lo (round 6: the sentinel draw 16 measured 2.8e-3 in that
        # tier with single bf16 P / V here after the UNet's flash kernel had been fixed)
        full = ops.attn_split()
        qk2 = ops.precise() and (getattr(self, "qk_split", False) or full)
        q = self.to_q.nhwc(g, out_dtype=ops.OUT_BF16, out_split=2 if qk2 else 1)
        k = self.to_k.nhwc(g, out_dtype=ops.OUT_BF16, out_split=2 if qk2 else 1)
        if qk2:
            k = ops.split_rows_hhl(k, Lp)
        elif Lp != L:
            kp = torch.zeros((N, Lp, Cc), device=x.device, dtype=ops.act_dtype())
            kp[:, :L] = k
            k = kp
        s = ops.bmm_nt(q, k, alpha=self.scale, out_dtype=ops.OUT_F32, both_split=qk2)         # [N, L, Lp] fp32 scores
        if full:
            vts = ops.transpose_split(self.to_v.nhwc(g, out_dtype=ops.OUT_F32), Lp)           # [N, 2C, Lp]: V^T hi rows, then lo rows
            vt = torch.cat([vts[:, :Cc], vts[:, :Cc], vts[:, Cc:]], dim=-1)                   # [N, C, 3 Lp] rows [v_hi | v_hi | v_lo] (plain device copies)
            p = ops.softmax_rows(s, valid=L, split=True)                                      # [N, L, 2 Lp] rows [p_hi | p_lo]
            del s, vts
            o = ops.bmm_nt(p, vt, out_split=self.to_out[0].in_split(), both_split=True)
        else:
            vt = ops.linear_t(g, self.to_v.packed(), L, ld=Lp)                    # [N, C, Lp], zero padded keys
            p = ops.softmax_rows(s, valid=L)
            del s
            o = ops.bmm_nt(p, vt, out_split=self.to_out[0].in_split())            # [N, L, C] operand of the output projection
        out = self.to_out[0].nhwc(o, residual=x.reshape(N, L, Cc), gn_groups=self.group_norm.num_groups)   # -> mid resnet norm1
        return ops.carry_gn(out, out.reshape(N, H, W, Cc))

    # ---- diffusers' Attention helper surface, as infer/vaehook.py:137-171 (attn_forward_new) calls it op by op on
    # [B, L, C] tensors in the CALLER's dtype (torch plumbing around the same GEMM / softmax kernels) -------------------
    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        return attention_mask                   # the VAE mid block never has a mask (the hook passes None through)

    def head_to_batch_dim(self, tensor, out_dim=3):
        B, L, Cc = tensor.shape
        t = tensor.reshape(B, L, self.heads, Cc // self.heads).permute(0, 2, 1, 3)
        return t.reshape(B * self.heads, L, Cc // self.heads) if out_dim == 3 else t

    def batch_to_head_dim(self, tensor):
        BH, L, d = tensor.shape
        return tensor.reshape(BH // self.heads, self.heads, L, d).permute(0, 2, 1, 3).reshape(BH // self.heads, L, d * self.heads)

    def get_attention_scores(self, query, key, attention_mask=None):
        """softmax(scale * Q K^T) [B*heads, Lq, Lk] in query's dtype: the batched MFMA GEMM (fp32 scores) + the masked row
        softmax (diffusers: baddbmm(beta=0, alpha=scale) -> float() softmax -> cast, upcast_softmax=True for the VAE)."""
        if attention_mask is not None:
            raise NotImplementedError("VaeAttention.get_attention_scores: the VAE mid block is never masked")
        B, Lq, d = query.shape
        Lk = key.shape[1]
        dp, Lkp = ops._round_up(d, 32), ops._round_up(Lk, 128)
        q = torch.zeros((B, Lq, dp), device=query.device, dtype=ops.act_dtype()); q[..., :d] = query
        k = torch.zeros((B, Lkp, dp), device=query.device, dtype=ops.act_dtype()); k[:, :Lk, :d] = key
        s = ops.bmm_nt(q, k, alpha=self.scale, out_dtype=ops.OUT_F32)
        return ops.softmax_rows(s, valid=Lk)[:, :, :Lk].to(query.dtype)


class _VaeMid(nn.Module):
    def __init__(self, ch: int, groups: int):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, groups, 1e-6), ResnetBlock2D(ch, ch, None, groups, 1e-6)])
        self.attentions = nn.ModuleList([VaeAttention(ch, groups)])

    def nhwc(self, h):
        h = self.resnets[0].nhwc(h)
        h = self.attentions[0].nhwc(h)
        return self.resnets[1].nhwc(h)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, None, groups, 1e-6) for j in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, 0)]) if add_down else None

    def nhwc(self, h):
        samp = self.downsamplers[0] if self.downsamplers is not None else None
        for j, r in enumerate(self.resnets):
            h = r.nhwc(h, out_for=samp.conv if (samp is not None and j == len(self.resnets) - 1) else None)
        if samp is not None:
            h = samp.nhwc(h, gn_groups=self.resnets[0].norm1.num_groups)
        return h


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, None, groups, 1e-6) for j in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def nhwc(self, h):
        samp = self.upsamplers[0] if self.upsamplers is not None else None
        for j, r in enumerate(self.resnets):
            h = r.nhwc(h, out_for=samp.conv if (samp is not None and j == len(self.resnets) - 1) else None)
        if samp is not None:
            h = samp.nhwc(h, gn_groups=self.resnets[0].norm1.num_groups)
        return h


class Encoder(nn.Module):
    def __init__(self, c: ConfigDict):
        super().__init__()
        boc, g = c.block_out_channels, c.norm_num_groups
        self.conv_in = Conv2d(c.in_channels, boc[0], 3, padding=1)
        blocks, out = [], boc[0]
        for i, ch in enumerate(boc):
            cin, out = out, ch
            blocks.append(DownEncoderBlock2D(cin, out, c.layers_per_block, g, add_down=i < len(boc) - 1))
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _VaeMid(boc[-1], g)
        self.conv_norm_out = GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(boc[-1], 2 * c.latent_channels, 3, padding=1)

    def nhwc(self, x):
        h = self.conv_in.nhwc(x, gn_groups=self.conv_norm_out.num_groups)
        for b in self.down_blocks:
            h = b.nhwc(h)
        h = self.mid_block.nhwc(h)
        return self.conv_out.nhwc(h, gn=self.conv_norm_out.spec(h, ops.ACT_SILU))

    def run_nhwc(self, x):
        """nhwc() or, when a tiled-VAE hook is installed (pipelines.vaehook.VAEHook), the hook."""
        hook = getattr(self, "_tile_hook", None)
        return hook(x) if hook is not None else self.nhwc(x)

    def forward(self, x):  # NCHW in/out (diffusers convention)
        y = self.run_nhwc(ops.nchw_to_nhwc(x.contiguous(), 8))
        # the reference's VAEHook returns an fp32 buffer whatever the net dtype (infer/vaehook.py:804, SURVEY C-10)
        dt = torch.float32 if getattr(self, "_tile_hook", None) is not None else _io_dtype(x)
        return ops.nhwc_to_nchw(y, channels=self.conv_out.out_channels, dtype=dt)


class Decoder(nn.Module):
    def __init__(self, c: ConfigDict):
        super().__init__()
        boc, g = c.block_out_channels, c.norm_num_groups
        rev = list(reversed(boc))
        self.conv_in = Conv2d(c.latent_channels, rev[0], 3, padding=1)
        self.mid_block = _VaeMid(rev[0], g)
        blocks, out = [], rev[0]
        for i, ch in enumerate(rev):
            cin, out = out, ch
            blocks.append(UpDecoderBlock2D(cin, out, c.layers_per_block + 1, g, add_up=i < len(boc) - 1))
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = GroupNorm(g, boc[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2d(boc[0], c.out_channels, 3, padding=1)

    def nhwc(self, z):
        h = self.conv_in.nhwc(z, gn_groups=self.conv_norm_out.num_groups)
        h = self.mid_block.nhwc(h)
        for b in self.up_blocks:
            h = b.nhwc(h)
        return self.conv_out.nhwc(h, gn=self.conv_norm_out.spec(h, ops.ACT_SILU))

    def run_nhwc(self, z):
        hook = getattr(self, "_tile_hook", None)
        return hook(z) if hook is not None else self.nhwc(z)

    def forward(self, z):
        y = self.run_nhwc(ops.nchw_to_nhwc(z.contiguous(), ops._round_up(z.shape[1], 8)))
        dt = torch.float32 if getattr(self, "_tile_hook", None) is not None else _io_dtype(z)
        return ops.nhwc_to_nchw(y, channels=self.conv_out.out_channels, dtype=dt)


def _io_dtype(x):
    return ops.io_dtype(x)


def _nchw_call(fn, x, channels):
    """Run an NHWC executor on an NCHW tensor (leaf modules called op by op by diffusers-style code)."""
    y = fn(ops.nchw_to_nhwc(x.contiguous(), ops._round_up(x.shape[1], 8)))
    return ops.nhwc_to_nchw(y, channels=channels, dtype=_io_dtype(x)).to(x.dtype)


class DiagonalGaussianDistribution:
    """Posterior handle returned by encode(); `sample()` draws eps from torch's global RNG on the
    device exactly like diffusers (SURVEY C-1) unless explicit noise was provided for parity runs."""

    def __init__(self, moments_nhwc: torch.Tensor, latent_channels: int, noise: Optional[torch.Tensor], out_dtype):
        self._m, self._c, self._noise, self._dt = moments_nhwc, latent_channels, noise, out_dtype

    def sample_nhwc(self, shift: float = 0.0, scale: float = 1.0, generator=None) -> torch.Tensor:
        N, h, w, _ = self._m.shape
        if self._noise is not None:
            eps = self._noise.to(device=self._m.device, dtype=torch.float32).permute(0, 2, 3, 1).contiguous()
        else:
            eps = torch.randn((N, h, w, self._c), device=self._m.device, dtype=torch.float32, generator=generator)
        return ops.vae_sample(self._m, eps, self._c, shift, scale)

    def sample(self, generator=None) -> torch.Tensor:
        z = self.sample_nhwc(generator=generator)
        return ops.nhwc_to_nchw(z, channels=self._c, dtype=self._dt)

    def mode(self) -> torch.Tensor:
        return ops.nhwc_to_nchw(self._m, channels=self._c, dtype=self._dt)


class AutoencoderKL(ModelMixin):
    config_name = "config.json"
    default_config = SD21_VAE_CONFIG

    def __init__(self, **cfg):
        super().__init__()
        c = ConfigDict({**SD21_VAE_CONFIG, **cfg})
        self.config = c
        self.encoder = Encoder(c)
        self.decoder = Decoder(c)
        self.quant_conv = Conv2d(2 * c.latent_channels, 2 * c.latent_channels, 1) if c.use_quant_conv else None
        self.post_quant_conv = Conv2d(c.latent_channels, c.latent_channels, 1) if c.use_post_quant_conv else None
        self.posterior_noise: Optional[torch.Tensor] = None   # explicit eps [N,C,h,w] for parity runs

    # ---- fast NHWC entry points used by the pipelines -------------------------------------
    def encode_moments_nhwc(self, x_nhwc8: torch.Tensor) -> torch.Tensor:
        m = self.encoder.run_nhwc(x_nhwc8)
        if self.quant_conv is not None:
            m = self.quant_conv.nhwc(m, pad=0)
        return m

    def decode_nhwc(self, z_nhwc: torch.Tensor) -> torch.Tensor:
        """z [N,h,w,C8] (latent channels first, zero padded to 8) -> image NHWC [N,8h,8w,8] (RGB + zero pad)."""
        if self.post_quant_conv is not None:
            z_nhwc = self.post_quant_conv.nhwc(z_nhwc, pad=0)
        return self.decoder.run_nhwc(z_nhwc)

    # ---- diffusers API ------------------------------------------------------------------
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        m = self.encode_moments_nhwc(ops.nchw_to_nhwc(x.contiguous(), 8))
        post = DiagonalGaussianDistribution(m, self.config.latent_channels, self.posterior_noise, _io_dtype(x))
        return SimpleNamespace(latent_dist=post) if return_dict else (post,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        img = self.decode_nhwc(ops.nchw_to_nhwc(z.contiguous(), ops._round_up(z.shape[1], 8)))
        out = ops.nhwc_to_nchw(img, channels=self.config.out_channels, dtype=_io_dtype(z))
        return SimpleNamespace(sample=out) if return_dict else (out,)

    def forward(self, sample, sample_posterior: bool = False):
        post = self.encode(sample).latent_dist
        z = post.sample() if sample_posterior else post.mode()
        return self.decode(z)

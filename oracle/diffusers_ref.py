"""ORACLE — test infrastructure only (never imported by omgsr_amd/).

CPU, pure-PyTorch, fp32 restatement of the diffusers==0.34.0 modules that OMGSR's inference hot
path calls (reference call sites: infer/omgsr_s_infer_model.py:11-15,75-85,173 and
infer/omgsr_f_infer_model.py:16,99-106,191-211).  diffusers itself is a third-party dependency
pinned in the reference's requirements.txt:4 and is NOT present under /root/reference nor
installed in this image, so its published algorithm is restated here from SURVEY.md Appendix A
with diffusers' module/attribute names, config fields and state-dict keys.

PARITY UNPINNED at the diffusers boundary: the reference ships no tests, golden vectors or
fixtures for these modules and the library cannot be imported here.  What IS pinned (tests/golden):
the reference's own pipeline/tiling/stitching/schedule code, captured by importing it.

Everything below is built from torch.nn primitives (Conv2d, GroupNorm, LayerNorm, Linear,
softmax) so each sub-block is self-evidently the textbook op.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class Config(dict):
    """dict with attribute access, like diffusers' FrozenDict."""
    __getattr__ = dict.__getitem__


# ----------------------------------------------------------------------------------------------
# shared blocks

def timestep_sinusoid(t: torch.Tensor, dim: int, flip_sin_to_cos: bool = True, freq_shift: float = 0.0,
                      max_period: int = 10000) -> torch.Tensor:
    """diffusers.models.embeddings.get_timestep_embedding: fp32, [cos | sin] when flipped."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """SURVEY A.1 'ResnetBlock2D(x, emb)'. temb_channels None => VAE variant."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: Optional[int], groups: int, eps: float):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.use_in_shortcut = in_channels != out_channels
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if self.use_in_shortcut else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, channels: int, padding: int):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:   # VAE: asymmetric pad then valid conv
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Attention(nn.Module):
    """diffusers Attention restricted to what the hot path uses.
    UNet: no group_norm, q/k/v without bias, out with bias.  VAE: group_norm, single head, all biased.
    Flux: biased, RMSNorm on q/k, optional added (context) projections, optional out-proj."""

    def __init__(self, query_dim: int, heads: int, dim_head: int, cross_attention_dim: Optional[int] = None,
                 bias: bool = False, norm_num_groups: Optional[int] = None, eps: float = 1e-5,
                 residual_connection: bool = False, qk_norm: bool = False, added_kv_proj_dim: Optional[int] = None,
                 pre_only: bool = False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.residual_connection = residual_connection
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps) if norm_num_groups else None
        kv_dim = cross_attention_dim or query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        if qk_norm:
            self.norm_q = RMSNorm(dim_head, eps)
            self.norm_k = RMSNorm(dim_head, eps)
        else:
            self.norm_q = self.norm_k = None
        if added_kv_proj_dim:
            self.add_q_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.add_k_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, inner, bias=True)
            self.norm_added_q = RMSNorm(dim_head, eps)
            self.norm_added_k = RMSNorm(dim_head, eps)
            self.to_add_out = nn.Linear(inner, query_dim, bias=True)
        else:
            self.add_q_proj = None
        self.to_out = None if pre_only else nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Identity()])

    # --- the diffusers Attention helper surface that infer/vaehook.py:137-171 calls op by op ---------
    norm_cross = None

    def prepare_attention_mask(self, attention_mask, target_length, batch_size):
        return attention_mask            # the hook only ever passes None

    def head_to_batch_dim(self, t):
        B, L, Cc = t.shape
        return t.reshape(B, L, self.heads, Cc // self.heads).permute(0, 2, 1, 3).reshape(B * self.heads, L, Cc // self.heads)

    def batch_to_head_dim(self, t):
        BH, L, d = t.shape
        return t.reshape(BH // self.heads, self.heads, L, d).permute(0, 2, 1, 3).reshape(BH // self.heads, L, d * self.heads)

    def get_attention_scores(self, query, key, attention_mask=None):
        return torch.softmax(torch.bmm(query, key.transpose(-1, -2)).float() * self.scale, dim=-1).to(query.dtype)

    def _heads(self, x):
        B, L, _ = x.shape
        return x.view(B, L, self.heads, self.dim_head).transpose(1, 2)

    @staticmethod
    def sdpa(q, k, v, scale):
        s = torch.matmul(q, k.transpose(-1, -2)) * scale
        return torch.matmul(s.softmax(dim=-1), v)

    def forward(self, hidden_states, encoder_hidden_states=None):
        """UNet / VAE path (AttnProcessor2_0)."""
        residual = hidden_states
        spatial = hidden_states.dim() == 4
        if spatial:
            B, Cc, H, W = hidden_states.shape
            hidden_states = hidden_states.view(B, Cc, H * W).transpose(1, 2)
        if self.group_norm is not None:
            hidden_states = self.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = self._heads(self.to_q(hidden_states)), self._heads(self.to_k(ctx)), self._heads(self.to_v(ctx))
        o = self.sdpa(q, k, v, self.scale).transpose(1, 2).reshape(hidden_states.shape[0], -1, self.heads * self.dim_head)
        o = self.to_out[0](o)
        if spatial:
            o = o.transpose(-1, -2).reshape(B, Cc, H, W)
        if self.residual_connection:
            o = o + residual
        return o


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        var = x.float().pow(2).mean(-1, keepdim=True)
        return x * torch.rsqrt(var + self.eps) * self.weight


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class GELUProj(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4, activation: str = "geglu"):
        super().__init__()
        inner = dim * mult
        act = GEGLU(dim, inner) if activation == "geglu" else GELUProj(dim, inner)
        self.net = nn.ModuleList([act, nn.Identity(), nn.Linear(inner, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_attention_dim=cross_attention_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim, activation="geglu")

    def forward(self, x, ehs):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ehs)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    """Linear-projection variant (use_linear_projection=True), one block."""

    def __init__(self, channels: int, heads: int, dim_head: int, cross_attention_dim: int, groups: int):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Linear(channels, channels)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Linear(channels, channels)

    def forward(self, x, ehs):
        B, Cc, H, W = x.shape
        res = x
        y = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, Cc)
        y = self.proj_in(y)
        for blk in self.transformer_blocks:
            y = blk(y, ehs)
        y = self.proj_out(y).reshape(B, H, W, Cc).permute(0, 3, 1, 2)
        return y + res


# ----------------------------------------------------------------------------------------------
# UNet2DConditionModel (SD2.1-base) — SURVEY A.1

SD21_UNET_CONFIG = dict(
    in_channels=4, out_channels=4, sample_size=64, block_out_channels=[320, 640, 1280, 1280],
    down_block_types=["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    up_block_types=["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    layers_per_block=2, attention_head_dim=[5, 10, 20, 20], cross_attention_dim=1024,
    use_linear_projection=True, norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0,
    downsample_padding=1)


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, layers, heads, cross_dim, groups, eps, add_down, down_pad, with_attn):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, temb, groups, eps) for j in range(layers)])
        if with_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, cross_dim, groups) for _ in range(layers)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(cout, down_pad)]) if add_down else None

    def forward(self, h, temb, ehs):
        outs = []
        for j, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.attentions is not None:
                h = self.attentions[j](h, ehs)
            outs.append(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs.append(h)
        return h, outs


class _UpBlock(nn.Module):
    def __init__(self, in_channels, prev_out, cout, temb, layers, heads, cross_dim, groups, eps, add_up, with_attn):
        super().__init__()
        rs = []
        for j in range(layers):
            skip = in_channels if j == layers - 1 else cout
            rin = prev_out if j == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(rs)
        if with_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, cross_dim, groups) for _ in range(layers)])
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, h, skips, temb, ehs):
        for j, r in enumerate(self.resnets):
            h = torch.cat([h, skips.pop()], dim=1)
            h = r(h, temb)
            if self.attentions is not None:
                h = self.attentions[j](h, ehs)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class _MidBlock(nn.Module):
    def __init__(self, ch, temb, heads, cross_dim, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, groups, eps), ResnetBlock2D(ch, ch, temb, groups, eps)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, ch // heads, cross_dim, groups)])

    def forward(self, h, temb, ehs):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, ehs)
        return self.resnets[1](h, temb)


class UNet2DConditionModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        c = Config({**SD21_UNET_CONFIG, **cfg})
        self.config = c
        boc = c.block_out_channels
        temb = boc[0] * 4
        heads = c.attention_head_dim if isinstance(c.attention_head_dim, (list, tuple)) else [c.attention_head_dim] * len(boc)
        g, eps = c.norm_num_groups, c.norm_eps
        self.conv_in = nn.Conv2d(c.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        downs, out_ch = [], boc[0]
        for i, t in enumerate(c.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            downs.append(_DownBlock(in_ch, out_ch, temb, c.layers_per_block, heads[i], c.cross_attention_dim, g, eps,
                                    add_down=i < len(boc) - 1, down_pad=c.downsample_padding,
                                    with_attn=t == "CrossAttnDownBlock2D"))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _MidBlock(boc[-1], temb, heads[-1], c.cross_attention_dim, g, eps)
        rev, rheads = list(reversed(boc)), list(reversed(heads))
        ups, out_ch = [], rev[0]
        for i, t in enumerate(c.up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            ups.append(_UpBlock(in_ch, prev, out_ch, temb, c.layers_per_block + 1, rheads[i], c.cross_attention_dim, g, eps,
                                add_up=i < len(boc) - 1, with_attn=t == "CrossAttnUpBlock2D"))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], c.out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states):
        B = sample.shape[0]
        t = torch.as_tensor([timestep], dtype=torch.int64).reshape(-1).expand(B)
        t_emb = timestep_sinusoid(t, self.config.block_out_channels[0], self.config.flip_sin_to_cos, self.config.freq_shift)
        emb = self.time_embedding(t_emb.to(sample.dtype))
        ehs = encoder_hidden_states
        if ehs.shape[0] != B:
            ehs = ehs.expand(B, -1, -1)
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, emb, ehs)
            skips += outs
        h = self.mid_block(h, emb, ehs)
        for blk in self.up_blocks:
            h = blk(h, skips, emb, ehs)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return SimpleNamespace(sample=h)


# ----------------------------------------------------------------------------------------------
# AutoencoderKL — SURVEY A.2

SD21_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                       latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, shift_factor=None,
                       use_quant_conv=True, use_post_quant_conv=True)
FLUX_VAE_CONFIG = dict(in_channels=3, out_channels=3, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                       latent_channels=16, norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159,
                       use_quant_conv=False, use_post_quant_conv=False)


class _VaeMid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, groups, 1e-6), ResnetBlock2D(ch, ch, None, groups, 1e-6)])
        self.attentions = nn.ModuleList([Attention(ch, 1, ch, bias=True, norm_num_groups=groups, eps=1e-6, residual_connection=True)])

    def forward(self, h):
        h = self.resnets[0](h)
        h = self.attentions[0](h)
        return self.resnets[1](h)


class _EncDown(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, None, groups, 1e-6) for j in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, 0)]) if add_down else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
        return h


class _DecUp(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, None, groups, 1e-6) for j in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class Encoder(nn.Module):
    def __init__(self, c: Config):
        super().__init__()
        boc, g = c.block_out_channels, c.norm_num_groups
        self.conv_in = nn.Conv2d(c.in_channels, boc[0], 3, padding=1)
        blocks, out = [], boc[0]
        for i, ch in enumerate(boc):
            cin, out = out, ch
            blocks.append(_EncDown(cin, out, c.layers_per_block, g, add_down=i < len(boc) - 1))
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _VaeMid(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * c.latent_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for b in self.down_blocks:
            h = b(h)
        h = self.mid_block(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class Decoder(nn.Module):
    def __init__(self, c: Config):
        super().__init__()
        boc, g = c.block_out_channels, c.norm_num_groups
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(c.latent_channels, rev[0], 3, padding=1)
        self.mid_block = _VaeMid(rev[0], g)
        blocks, out = [], rev[0]
        for i, ch in enumerate(rev):
            cin, out = out, ch
            blocks.append(_DecUp(cin, out, c.layers_per_block + 1, g, add_up=i < len(boc) - 1))
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], c.out_channels, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid_block(h)
        for b in self.up_blocks:
            h = b(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class DiagonalGaussianDistribution:
    def __init__(self, moments: torch.Tensor, noise: Optional[torch.Tensor] = None):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self._noise = noise

    def sample(self, generator=None):
        eps = self._noise if self._noise is not None else torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + self.std * eps


class AutoencoderKL(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        c = Config({**SD21_VAE_CONFIG, **cfg})
        self.config = c
        self.encoder = Encoder(c)
        self.decoder = Decoder(c)
        self.quant_conv = nn.Conv2d(2 * c.latent_channels, 2 * c.latent_channels, 1) if c.use_quant_conv else None
        self.post_quant_conv = nn.Conv2d(c.latent_channels, c.latent_channels, 1) if c.use_post_quant_conv else None
        self.posterior_noise: Optional[torch.Tensor] = None   # explicit eps (SURVEY C-1): set by the harness

    @property
    def dtype(self):
        return self.encoder.conv_in.weight.dtype

    def encode(self, x):
        m = self.encoder(x)
        if self.quant_conv is not None:
            m = self.quant_conv(m)
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(m, self.posterior_noise))

    def decode(self, z, return_dict: bool = True):
        if self.post_quant_conv is not None:
            z = self.post_quant_conv(z)
        img = self.decoder(z)
        return SimpleNamespace(sample=img) if return_dict else (img,)


# ----------------------------------------------------------------------------------------------
# DDPMScheduler — SURVEY A.4

class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)


# ----------------------------------------------------------------------------------------------
# FluxTransformer2DModel — SURVEY A.3

FLUX_DEV_CONFIG = dict(patch_size=1, in_channels=64, out_channels=None, num_layers=19, num_single_layers=38,
                       attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096,
                       pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=[16, 56, 56])


def rope_1d(dim: int, pos: torch.Tensor, theta: float = 10000.0):
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64) / dim))
    ang = torch.outer(pos.to(torch.float64), freqs)
    return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()


def flux_pos_embed(ids: torch.Tensor, axes_dim):
    cos, sin = [], []
    pos = ids.float()
    for i, d in enumerate(axes_dim):
        c, s = rope_1d(d, pos[:, i])
        cos.append(c); sin.append(s)
    return torch.cat(cos, dim=-1), torch.cat(sin, dim=-1)


def apply_rotary_emb(x, cos, sin):
    """x [B, H, L, D]; interleaved real pairs (use_real_unbind_dim=-1)."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


def _joint_attention(attn: Attention, x, ctx, rope):
    B = x.shape[0]
    q, k, v = attn._heads(attn.to_q(x)), attn._heads(attn.to_k(x)), attn._heads(attn.to_v(x))
    q, k = attn.norm_q(q), attn.norm_k(k)
    n_ctx = 0
    if ctx is not None:
        cq, ck, cv = attn._heads(attn.add_q_proj(ctx)), attn._heads(attn.add_k_proj(ctx)), attn._heads(attn.add_v_proj(ctx))
        cq, ck = attn.norm_added_q(cq), attn.norm_added_k(ck)
        q, k, v = torch.cat([cq, q], dim=2), torch.cat([ck, k], dim=2), torch.cat([cv, v], dim=2)
        n_ctx = ctx.shape[1]
    if rope is not None:
        q, k = apply_rotary_emb(q, *rope), apply_rotary_emb(k, *rope)
    o = Attention.sdpa(q, k, v, attn.scale).transpose(1, 2).reshape(B, -1, attn.heads * attn.dim_head)
    if ctx is not None:
        return attn.to_out[0](o[:, n_ctx:]), attn.to_add_out(o[:, :n_ctx])
    return o


class AdaLayerNormZero(nn.Module):
    def __init__(self, dim: int, chunks: int):
        super().__init__()
        self.chunks = chunks
        self.linear = nn.Linear(dim, chunks * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb):
        parts = self.linear(F.silu(emb)).chunk(self.chunks, dim=1)
        shift, scale = parts[0], parts[1]
        return (self.norm(x) * (1 + scale[:, None]) + shift[:, None], *parts[2:])


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim):
        super().__init__()
        self.norm1 = AdaLayerNormZero(dim, 6)
        self.norm1_context = AdaLayerNormZero(dim, 6)
        self.attn = Attention(dim, heads, head_dim, bias=True, eps=1e-6, qk_norm=True, added_kv_proj_dim=dim)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim, activation="gelu-approximate")
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = FeedForward(dim, activation="gelu-approximate")

    def forward(self, h, c, temb, rope):
        hn, gate_a, shift_m, scale_m, gate_m = self.norm1(h, temb)
        cn, c_gate_a, c_shift_m, c_scale_m, c_gate_m = self.norm1_context(c, temb)
        h_attn, c_attn = _joint_attention(self.attn, hn, cn, rope)
        h = h + gate_a[:, None] * h_attn
        h = h + gate_m[:, None] * self.ff(self.norm2(h) * (1 + scale_m[:, None]) + shift_m[:, None])
        c = c + c_gate_a[:, None] * c_attn
        c = c + c_gate_m[:, None] * self.ff_context(self.norm2_context(c) * (1 + c_scale_m[:, None]) + c_shift_m[:, None])
        return c, h


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, mlp_ratio=4.0):
        super().__init__()
        self.mlp_hidden = int(dim * mlp_ratio)
        self.norm = AdaLayerNormZero(dim, 3)
        self.proj_mlp = nn.Linear(dim, self.mlp_hidden)
        self.proj_out = nn.Linear(dim + self.mlp_hidden, dim)
        self.attn = Attention(dim, heads, head_dim, bias=True, eps=1e-6, qk_norm=True, pre_only=True)

    def forward(self, x, temb, rope):
        xn, gate = self.norm(x, temb)
        mlp = F.gelu(self.proj_mlp(xn), approximate="tanh")
        a = _joint_attention(self.attn, xn, None, rope)
        return x + gate[:, None] * self.proj_out(torch.cat([a, mlp], dim=2))


class _TextProj(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class _TimeTextEmbed(nn.Module):
    def __init__(self, dim, pooled_dim, guidance: bool):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, dim)
        self.guidance_embedder = TimestepEmbedding(256, dim) if guidance else None
        self.text_embedder = _TextProj(pooled_dim, dim)

    def forward(self, timestep, guidance, pooled):
        e = self.timestep_embedder(timestep_sinusoid(timestep, 256).to(pooled.dtype))
        if self.guidance_embedder is not None:
            e = e + self.guidance_embedder(timestep_sinusoid(guidance, 256).to(pooled.dtype))
        return e + self.text_embedder(pooled)


class AdaLayerNormContinuous(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, cond):
        scale, shift = self.linear(F.silu(cond)).chunk(2, dim=1)   # scale FIRST
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class FluxTransformer2DModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        c = Config({**FLUX_DEV_CONFIG, **cfg})
        self.config = c
        dim = c.num_attention_heads * c.attention_head_dim
        self.inner_dim = dim
        self.x_embedder = nn.Linear(c.in_channels, dim)
        self.context_embedder = nn.Linear(c.joint_attention_dim, dim)
        self.time_text_embed = _TimeTextEmbed(dim, c.pooled_projection_dim, c.guidance_embeds)
        self.transformer_blocks = nn.ModuleList([FluxTransformerBlock(dim, c.num_attention_heads, c.attention_head_dim) for _ in range(c.num_layers)])
        self.single_transformer_blocks = nn.ModuleList([FluxSingleTransformerBlock(dim, c.num_attention_heads, c.attention_head_dim) for _ in range(c.num_single_layers)])
        self.norm_out = AdaLayerNormContinuous(dim)
        self.proj_out = nn.Linear(dim, c.patch_size * c.patch_size * (c.out_channels or c.in_channels))

    @property
    def dtype(self):
        return self.x_embedder.weight.dtype

    def forward(self, hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids,
                return_dict: bool = True):
        h = self.x_embedder(hidden_states)
        B = h.shape[0]
        timestep = timestep.to(h.dtype) * 1000
        guidance = guidance.to(h.dtype) * 1000 if guidance is not None else None
        temb = self.time_text_embed(timestep, guidance, pooled_projections)
        if temb.shape[0] != B:
            temb = temb.expand(B, -1)
        c = self.context_embedder(encoder_hidden_states)
        if c.shape[0] != B:
            c = c.expand(B, -1, -1)
        ids = torch.cat([txt_ids, img_ids], dim=0)
        rope = flux_pos_embed(ids, self.config.axes_dims_rope)
        for blk in self.transformer_blocks:
            c, h = blk(h, c, temb, rope)
        n_txt = c.shape[1]
        x = torch.cat([c, h], dim=1)
        for blk in self.single_transformer_blocks:
            x = blk(x, temb, rope)
        h = x[:, n_txt:]
        out = self.proj_out(self.norm_out(h, temb))
        return SimpleNamespace(sample=out) if return_dict else (out,)


# ----------------------------------------------------------------------------------------------
# PEFT LoRA merge: W += (alpha / r) * B @ A, conv: contract r.

def merge_lora_(module: nn.Module, adapter_sd: dict, r: int, lora_alpha: float, prefix: str = "base_model.model.") -> int:
    scale = lora_alpha / r
    merged = 0
    mods = dict(module.named_modules())
    for key, A in adapter_sd.items():
        if not key.endswith("lora_A.weight"):
            continue
        name = key[len(prefix):-len(".lora_A.weight")] if key.startswith(prefix) else key[:-len(".lora_A.weight")]
        Bm = adapter_sd[key.replace("lora_A", "lora_B")]
        tgt = mods[name]
        with torch.no_grad():
            if A.dim() == 4:
                delta = torch.einsum("or,rikl->oikl", Bm[:, :, 0, 0].float(), A.float())
            else:
                delta = Bm.float() @ A.float()
            tgt.weight += (scale * delta).to(tgt.weight.dtype)
        merged += 1
    return merged

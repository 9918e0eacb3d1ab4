"""FluxTransformer2DModel (FLUX.1-dev MM-DiT) with diffusers' API surface on the gfx950 kernels.

Stands in for `diffusers.FluxTransformer2DModel` at infer/omgsr_f_infer_model.py:103-106,191-200,271-280:
keyword call `flux(hidden_states=, timestep=, guidance=, pooled_projections=, encoder_hidden_states=,
txt_ids=, img_ids=, return_dict=False)[0]`, `.dtype`. State-dict keys: SURVEY A.5.

MI355X-first execution:
  * OMGSR calls the DiT at ONE sigma(t*), ONE guidance value and ONE pooled prompt: `temb` is a constant,
    so every AdaLN-Zero modulation vector (3.28 B parameters that only ever see that 1-token constant:
    norm1/norm1_context/norm.linear/norm_out.linear and the three embedder MLPs) is folded once, in fp32,
    into per-block (1+scale, shift, gate) vectors. Per image the blocks then run only the token GEMMs.
  * LayerNorm+modulate is one row kernel (a[c]*x^+b[c]); gate*y+residual, GELU-tanh and biases are GEMM
    epilogues; q|k come out of ONE fused projection GEMM per stream, written straight into the joint
    [text ; image] sequence buffer; V is written transposed into the joint V^T buffer
  * RMSNorm(q,k)*w + RoPE is one in-place pass over the fused [q|k] buffer (per-head weight table)
  * joint attention (24 heads x 128, 4608 keys) is the fused MFMA attention kernel; in the single blocks it
    writes directly into the [attn | mlp] concat buffer that proj_out consumes
Batch: the B images of a call ride along M in every launch (the reference is batch-1, infer/infer_omgsr_f.py:95; images
are independent): the single-stream blocks (2/3 of the FLOPs) run GEMMs of B * 4608 rows, so the 24 GB of weights are read
once per call instead of once per image and the 360-432-tile grids of one sequence stop quantising on the 512 workgroup
slots; the double-stream blocks' projections into / out of the joint [text ; image] buffers run as B GEMMs in one grid.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..nn import InputCache, LayerNorm, Linear, RMSNormWeight, _key, bump_cache_epoch
from .modeling_utils import ConfigDict, ModelMixin
from .unet_2d_condition import TimestepEmbedding, timestep_sinusoid

FLUX_DEV_CONFIG = dict(patch_size=1, in_channels=64, out_channels=None, num_layers=19, num_single_layers=38,
                       attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096,
                       pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=[16, 56, 56])


class _Cached:
    def _cache(self, name, builder, *tensors):
        k = _key(*tensors)
        slot = self.__dict__.setdefault("_pk_cache", {})
        if name not in slot or slot[name][0] != k:
            bump_cache_epoch()
            slot[name] = (k, builder())
        return slot[name][1]


class FluxAttention(nn.Module, _Cached):
    def __init__(self, dim: int, heads: int, head_dim: int, joint: bool, pre_only: bool):
        super().__init__()
        inner = heads * head_dim
        self.heads, self.head_dim, self.inner, self.scale = heads, head_dim, inner, head_dim ** -0.5
        self.to_q, self.to_k, self.to_v = Linear(dim, inner), Linear(dim, inner), Linear(dim, inner)
        self.norm_q, self.norm_k = RMSNormWeight(head_dim, 1e-6), RMSNormWeight(head_dim, 1e-6)
        if joint:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = Linear(dim, inner), Linear(dim, inner), Linear(dim, inner)
            self.norm_added_q, self.norm_added_k = RMSNormWeight(head_dim, 1e-6), RMSNormWeight(head_dim, 1e-6)
            self.to_add_out = Linear(inner, dim)
        self.to_out = None if pre_only else nn.ModuleList([Linear(inner, dim), nn.Dropout(0.0)])

    def qk_packed(self, ctx: bool = False) -> ops.PackedWeight:
        q, k = (self.add_q_proj, self.add_k_proj) if ctx else (self.to_q, self.to_k)
        sp, wsp = q.in_split(), q.in_wsplit()
        return self._cache("qk_ctx" if ctx else "qk", lambda: ops.pack_linear_weight(
            torch.cat([q.weight, k.weight], 0), torch.cat([q.bias, k.bias], 0), split=sp, w_split=wsp), q.weight, k.weight, q.bias, k.bias, sp, wsp)

    def norm_table(self, ctx: bool = False) -> torch.Tensor:
        nq, nk = (self.norm_added_q, self.norm_added_k) if ctx else (self.norm_q, self.norm_k)
        return self._cache("nt_ctx" if ctx else "nt", lambda: torch.cat(
            [nq.w32()[None].expand(self.heads, -1), nk.w32()[None].expand(self.heads, -1)], 0).contiguous(), nq.weight, nk.weight)


class GELUProj(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = Linear(dim_in, dim_out)


class FluxFeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GELUProj(dim, dim * mult), nn.Dropout(0.0), Linear(dim * mult, dim)])

    def run(self, xn, residual, gate):
        h = self.net[0].proj.nhwc(xn, act=ops.ACT_GELU_TANH, out_dtype=ops.OUT_BF16, out_split=self.net[2].in_split())
        return self.net[2].nhwc(h, residual=residual, gate=gate)


class AdaLayerNormZero(nn.Module):
    def __init__(self, dim: int, chunks: int):
        super().__init__()
        self.chunks = chunks
        self.linear = Linear(dim, chunks * dim)
        self.norm = LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    @torch.no_grad()
    def fold(self, temb32: torch.Tensor):
        """temb [dim] fp32 -> `chunks` fp32 vectors linear(silu(temb)) (constant folding, once per (t*, guidance, prompt))."""
        m = ops.linear_f32(temb32, self.linear.weight, self.linear.bias, silu_in=True)
        return [c.contiguous() for c in m.chunk(self.chunks)]


class FluxTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim):
        super().__init__()
        self.norm1 = AdaLayerNormZero(dim, 6)
        self.norm1_context = AdaLayerNormZero(dim, 6)
        self.attn = FluxAttention(dim, heads, head_dim, joint=True, pre_only=False)
        self.norm2 = LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FluxFeedForward(dim)
        self.norm2_context = LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = FluxFeedForward(dim)

    def fold(self, temb32):
        def six(norm):
            shift_a, scale_a, gate_a, shift_m, scale_m, gate_m = norm.fold(temb32)
            return dict(a1=(1 + scale_a).contiguous(), b1=shift_a, g1=gate_a, a2=(1 + scale_m).contiguous(), b2=shift_m, g2=gate_m)
        return dict(img=six(self.norm1), ctx=six(self.norm1_context))

    def run(self, h, c, mod, rope, ws):
        """h [B, Li, D] image tokens, c [B, Lc, D] text tokens (stream tensors); ws = joint operand buffers [B, L, ...].
        All B images ride in every launch: row-wise kernels see B*L rows, the projections into / out of the joint buffers run as
        B GEMMs of one grid (grid.z = B, one weight read), attention takes the batch stride."""
        at = self.attn
        Lc, Li = c.shape[1], h.shape[1]
        mi, mc = mod["img"], mod["ctx"]
        hn = ops.layer_norm(h, mi["a1"], mi["b1"], 1e-6, split=at.to_q.in_split())
        cn = ops.layer_norm(c, mc["a1"], mc["b1"], 1e-6, split=at.add_q_proj.in_split())
        osp = at.to_out[0].in_split()                      # to_out and to_add_out read one buffer: same form (precision.check_policy)
        qk, vt, o = ws["qk"], ws["vt"], ws["o2" if osp == 2 else "o"]
        ops.linear_into(cn, at.qk_packed(ctx=True), qk, 0, 0)
        ops.linear_into(hn, at.qk_packed(), qk, Lc, 0)
        ops.linear_t_into(cn, at.add_v_proj.packed(), vt, 0)
        ops.linear_t_into(hn, at.to_v.packed(), vt, Lc)
        cos, sin = rope
        ops.rmsnorm_rope_(qk, at.norm_table(ctx=True), cos, sin, 2 * at.heads, at.head_dim, pos0=0, w_after=at.norm_table(), split_at=Lc)
        ops.attention(qk, qk, vt, at.heads, at.head_dim, at.scale, q_col=0, k_col=at.inner, Lk=Lc + Li, out=o, out_split=osp)
        h = ops.linear_rows(o, Lc, Li, at.to_out[0].packed(), residual=h, gate=mi["g1"])
        c = ops.linear_rows(o, 0, Lc, at.to_add_out.packed(), residual=c, gate=mc["g1"])
        h = self.ff.run(ops.layer_norm(h, mi["a2"], mi["b2"], 1e-6, split=self.ff.net[0].proj.in_split()), h, mi["g2"])
        c = self.ff_context.run(ops.layer_norm(c, mc["a2"], mc["b2"], 1e-6, split=self.ff_context.net[0].proj.in_split()), c, mc["g2"])
        return h, c


class FluxSingleTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, mlp_ratio=4.0):
        super().__init__()
        self.mlp_hidden = int(dim * mlp_ratio)
        self.norm = AdaLayerNormZero(dim, 3)
        self.proj_mlp = Linear(dim, self.mlp_hidden)
        self.act_mlp = nn.GELU(approximate="tanh")
        self.proj_out = Linear(dim + self.mlp_hidden, dim)
        self.attn = FluxAttention(dim, heads, head_dim, joint=False, pre_only=True)
        self.attn._shares_input_with = (self.proj_mlp,)        # precision.check_policy: one LayerNorm'd operand feeds all four

    def fold(self, temb32):
        shift, scale, gate = self.norm.fold(temb32)
        return dict(a=(1 + scale).contiguous(), b=shift, g=gate)

    def run(self, x, mod, rope, ws):
        """x [B, L, D] joint [text ; image] tokens (stream tensor): every GEMM sees M = B * L rows."""
        at = self.attn
        B, L, D = x.shape
        xn = ops.layer_norm(x, mod["a"], mod["b"], 1e-6, split=at.to_q.in_split())      # read by to_q | to_k, to_v and proj_mlp
        csp = self.proj_out.in_split()
        qk, vt, cat = ws["qk"], ws["vt"], ws["cat2" if csp == 2 else "cat"]
        Kc = D + self.mlp_hidden                           # the [attn | mlp] operand of proj_out; split form: [hi (Kc) | lo (Kc)]
        ops.linear_into(xn.reshape(B * L, -1), at.qk_packed(), qk.reshape(B * L, -1), 0, 0, sample_rows=L)
        ops.linear_t_into(xn, at.to_v.packed(), vt, 0)
        ops.rmsnorm_rope_(qk, at.norm_table(), rope[0], rope[1], 2 * at.heads, at.head_dim, pos0=0)
        ops.attention(qk, qk, vt, at.heads, at.head_dim, at.scale, q_col=0, k_col=at.inner, Lk=L, out=cat,
                      out_split=csp, o_lo_col=Kc)
        cat2d = cat.reshape(B * L, -1)
        ops.linear_into(xn.reshape(B * L, -1), self.proj_mlp.packed(), cat2d, 0, D, act=ops.ACT_GELU_TANH, out_split=csp, lo_col0=Kc + D, sample_rows=L)
        return self.proj_out.nhwc(cat, residual=x, gate=mod["g"])


class _TextProj(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = Linear(in_dim, dim)
        self.act_1 = nn.SiLU()
        self.linear_2 = Linear(dim, dim)

    def fp32(self, x):
        h = ops.linear_f32(x, self.linear_1.weight, self.linear_1.bias)
        return ops.linear_f32(h, self.linear_2.weight, self.linear_2.bias, silu_in=True)


class _TimeTextEmbed(nn.Module):
    def __init__(self, dim, pooled_dim, guidance: bool):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, dim)
        self.guidance_embedder = TimestepEmbedding(256, dim) if guidance else None
        self.text_embedder = _TextProj(pooled_dim, dim)

    @torch.no_grad()
    def fp32(self, timestep, guidance, pooled):
        e = self.timestep_embedder.fp32(timestep_sinusoid(timestep, 256, True, 0.0))
        if self.guidance_embedder is not None:
            e = e + self.guidance_embedder.fp32(timestep_sinusoid(guidance, 256, True, 0.0))
        return e + self.text_embedder.fp32(pooled.float())


class AdaLayerNormContinuous(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear = Linear(dim, 2 * dim)
        self.norm = LayerNorm(dim, elementwise_affine=False, eps=1e-6)


def rope_tables(ids: torch.Tensor, axes_dim, theta: float = 10000.0):
    """FluxPosEmbed: per axis float64 angles, cos/sin repeat-interleaved (real pairs) -> fp32 [L, sum(axes)]."""
    cos, sin = [], []
    pos = ids.to(torch.float64)
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64, device=ids.device) / d))
        ang = torch.outer(pos[:, i], freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, -1).contiguous(), torch.cat(sin, -1).contiguous()


class FluxTransformer2DModel(ModelMixin, _Cached):
    default_config = FLUX_DEV_CONFIG

    def __init__(self, **cfg):
        super().__init__()
        c = ConfigDict({**FLUX_DEV_CONFIG, **cfg})
        self.config = c
        dim = c.num_attention_heads * c.attention_head_dim
        self.inner_dim = dim
        if sum(c.axes_dims_rope) != c.attention_head_dim or c.attention_head_dim != 128:
            raise ValueError("the gfx950 Flux path is specialised for head_dim 128 = sum(axes_dims_rope)")
        self.x_embedder = Linear(c.in_channels, dim)
        self.context_embedder = Linear(c.joint_attention_dim, dim)
        self.time_text_embed = _TimeTextEmbed(dim, c.pooled_projection_dim, c.guidance_embeds)
        self.transformer_blocks = nn.ModuleList([FluxTransformerBlock(dim, c.num_attention_heads, c.attention_head_dim) for _ in range(c.num_layers)])
        self.single_transformer_blocks = nn.ModuleList([FluxSingleTransformerBlock(dim, c.num_attention_heads, c.attention_head_dim) for _ in range(c.num_single_layers)])
        self.norm_out = AdaLayerNormContinuous(dim)
        self.proj_out = Linear(dim, c.patch_size * c.patch_size * (c.out_channels or c.in_channels))
        self._mod_cache, self._rope_cache, self._ctx_cache = InputCache(), InputCache(), InputCache()
        # 16-bit weights: round timestep / guidance through the weight dtype before the x1000 like diffusers does (the LoRA was
        # trained through that path); set False to condition on the exact value (what an fp32 model sees)
        self.round_timestep_to_weight_dtype = True

    # ---- constant folding ------------------------------------------------------------------
    def _host_scalar(self, slot: str, t: Optional[torch.Tensor]):
        """First element of a (device) tensor as a Python float, read ONCE per tensor object and version: OMGSR calls the DiT with the
        same timestep / guidance tensors on every image (the pipelines keep them), so the device -> host read (a stream
        synchronisation) happens on the first call only."""
        if t is None:
            return None
        hit = self.__dict__.get("_scalar_" + slot)
        if hit is not None and hit[0] is t and hit[1] == t._version:
            return hit[2]
        v = float(t.reshape(-1)[0].float())
        self.__dict__["_scalar_" + slot] = (t, t._version, v)
        return v

    def _mod_deps(self):
        """The parameters the modulation tables depend on, as a flat cached list. load_state_dict / .to() / in-place LoRA merges keep
        the parameter OBJECTS (and bump _version), but load_state_dict(assign=True) / swap_tensors-style conversion REPLACE them: the
        cache holds (owner module, name, parameter) and is rebuilt when any owner no longer holds that object (360 `is` tests per
        forward instead of a recursive .parameters() walk)."""
        owners = self.__dict__.get("_mod_dep_owners")
        if owners is not None and all(m._parameters.get(n) is p for m, n, p in owners):
            return self.__dict__["_mod_dep_list"]
        mods = [self.time_text_embed, self.norm_out]
        for b in list(self.transformer_blocks) + list(self.single_transformer_blocks):
            mods += [getattr(b, nm) for nm in ("norm1", "norm1_context", "norm") if hasattr(b, nm)]
        owners = [(sub, n, p) for m in mods for sub in m.modules() for n, p in sub._parameters.items() if p is not None]
        self.__dict__["_mod_dep_owners"] = owners
        self.__dict__["_mod_dep_list"] = [p for _, _, p in owners]
        return self.__dict__["_mod_dep_list"]

    def _mod_key(self, timestep, guidance) -> tuple:
        """(scaled timestep, scaled guidance, identity of every parameter the modulation tables depend on)."""
        wd = self.x_embedder.weight.dtype
        rt = (lambda v: float((torch.tensor(v, dtype=torch.float32).to(wd) * 1000).float())) if (wd != torch.float32 and self.round_timestep_to_weight_dtype) \
            else (lambda v: v * 1000.0)
        t = rt(self._host_scalar("t", timestep))
        g = None if guidance is None else rt(self._host_scalar("g", guidance))
        # address + version + dtype of ~360 parameters: the version bump of an in-place LoRA merge or a .to() re-keys the tables
        return (t, g, ops.mode_key(), tuple((p.data_ptr(), p._version, p.dtype) for p in self._mod_deps()))

    @torch.no_grad()
    def _modulation(self, timestep: torch.Tensor, guidance: Optional[torch.Tensor], pooled: torch.Tensor):
        # diffusers: `timestep.to(hidden_states.dtype) * 1000` - a 16-bit model sees the timestep ROUNDED to its dtype before
        # the scaling (bf16: 0.50511 -> 0.50390625 -> 503.9 -> bf16 504.0; SURVEY C-7), the fp32 model 505.11
        wkey = self._mod_key(timestep, guidance)
        t, g = wkey[0], wkey[1]

        def build():
            dev = self.x_embedder.weight.device
            tt = torch.tensor([t], dtype=torch.float32, device=dev)
            gg = None if g is None else torch.tensor([g], dtype=torch.float32, device=dev)
            temb = self.time_text_embed.fp32(tt, gg, pooled.to(dev)[:1])[0]
            double = [b.fold(temb) for b in self.transformer_blocks]
            single = [b.fold(temb) for b in self.single_transformer_blocks]
            scale, shift = ops.linear_f32(temb, self.norm_out.linear.weight, self.norm_out.linear.bias, silu_in=True).chunk(2)   # scale FIRST
            out = dict(a=(1 + scale).contiguous(), b=shift.contiguous())
            return dict(double=double, single=single, out=out)
        return self._mod_cache.get((pooled,), wkey, build)

    def _rope(self, txt_ids, img_ids):
        # keyed on the id TENSORS (identity + version, references held): an address alone is not an identity
        return self._rope_cache.get((txt_ids, img_ids), (), lambda: rope_tables(torch.cat([txt_ids, img_ids], 0).float(), self.config.axes_dims_rope))

    def _context(self, ehs):
        def build():
            e = ehs.float().contiguous() if ops.precise() else ehs.to(ops.act_dtype()).contiguous()
            return self.context_embedder.nhwc(e)
        return self._ctx_cache.get((ehs,), self._ctx_key(), build)

    def _ctx_key(self) -> tuple:
        return _key(self.context_embedder.weight, self.context_embedder.bias, self.context_embedder.in_split(), self.context_embedder.in_wsplit())

    # ---- token executor --------------------------------------------------------------------
    def tokens(self, x_tok: torch.Tensor, timestep, guidance, pooled, ehs, txt_ids, img_ids) -> torch.Tensor:
        """x_tok [B, Li, in_channels] stream tensor -> velocity [B, Li, in_channels] stream tensor."""
        mod = self._modulation(timestep, guidance, pooled)
        rope = self._rope(txt_ids, img_ids)
        ctx0 = self._context(ehs)                               # [1|B, Lc, D]
        B, Li, _ = x_tok.shape
        Lc, D = ctx0.shape[1], self.inner_dim
        L = Lc + Li
        dev = x_tok.device
        mlp = self.single_transformer_blocks[0].mlp_hidden if len(self.single_transformer_blocks) else 0
        ad = ops.act_dtype()
        ws = dict(qk=torch.empty((B, L, 2 * D), device=dev, dtype=ad),
                  vt=torch.empty((B, D, ops._round_up(L, 8)), device=dev, dtype=ad),
                  o=torch.empty((B, L, D), device=dev, dtype=ad),
                  cat=torch.empty((B, L, D + mlp), device=dev, dtype=ad))
        if ops.precise():       # two-term split forms of the operands that to_out / to_add_out / proj_out read (policy dependent)
            if any(b.attn.to_out[0].in_split() == 2 for b in self.transformer_blocks):
                ws["o2"] = torch.empty((B, L, 2 * D), device=dev, dtype=ad)
            if any(b.proj_out.in_split() == 2 for b in self.single_transformer_blocks):
                ws["cat2"] = torch.empty((B, L, 2 * (D + mlp)), device=dev, dtype=ad)
        if ws["vt"].shape[-1] != L:
            ws["vt"].zero_()
        h = self.x_embedder.nhwc(x_tok)                                                   # [B, Li, D]
        c = ctx0.expand(B, -1, -1).contiguous() if ctx0.shape[0] != B else ctx0           # text stream: per image after block 1
        for blk, m in zip(self.transformer_blocks, mod["double"]):
            h, c = blk.run(h, c, m, rope, ws)
        x = torch.cat([c, h], 1)                                                          # [B, L, D] (plumbing copy, once)
        for blk, m in zip(self.single_transformer_blocks, mod["single"]):
            x = blk.run(x, m, rope, ws)
        hn = ops.layer_norm(x[:, Lc:].contiguous(), mod["out"]["a"], mod["out"]["b"], 1e-6, split=self.proj_out.in_split())
        return self.proj_out.nhwc(hn)

    # ---- diffusers API ---------------------------------------------------------------------
    def forward(self, hidden_states, timestep, guidance=None, pooled_projections=None, encoder_hidden_states=None,
                txt_ids=None, img_ids=None, return_dict: bool = True, **_):
        x = hidden_states.to(ops.stream_dtype()).contiguous()
        out = self.tokens(x, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids)
        out = out.to(hidden_states.dtype)
        return SimpleNamespace(sample=out) if return_dict else (out,)

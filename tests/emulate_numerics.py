"""CPU emulation of where the HIP path rounds (test infrastructure, not collected by pytest).

Runs the fp32 oracle modules (oracle/diffusers_ref.py) with explicit rounding points q_*() inserted where a
storage / operand scheme of the GPU path would round, and reports rel-L2 / PSNR of the emulated OMGSR-S output
against the un-rounded fp32 oracle at FULL SD2.1 shapes. Used to choose the accuracy tier before writing kernels:

    python tests/emulate_numerics.py [--side 512] [--schemes all16,stream32,inner32]

Rounding classes:
  operand  every tensor that feeds an MFMA (conv / linear inputs, q, k, v, softmax probabilities)
  inner    a conv / linear output that is only normalised and fed to the next conv (resnet conv1 output, FF hidden ...)
  stream   the residual stream (resnet outputs, transformer residual sums, skip tensors)
"""
from __future__ import annotations

import argparse
import os
import re
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq  # noqa: E402
from oracle import diffusers_ref as R  # noqa: E402
from oracle.pipeline_ref import OmgsrSRef, stitch, tile_offsets  # noqa: E402


class Scheme:
    """only: optional set of (stage, kind) pairs whose OPERAND roundings are kept (all others exact) - error budget runs.
    stage in {enc, unet, dec, lat}; kind in {conv, lin, attn, lat}."""

    def __init__(self, operand, inner, stream, only=None, hilo=None, hilo_fn=None, names=None, act_split=None, inner16=None, qk=None):
        """names / act_split: name-based policy (the product's own form, omgsr_amd/precision.py): names maps id(module) ->
        (model key, qualified name), act_split maps model key -> list of regular expressions; an operand whose CONSUMER module
        matches is carried as the two-term split. Operands without a consumer module (latents) count as exact."""
        self.operand, self.inner, self.stream = operand, inner, stream
        self.only, self.hilo, self.hilo_fn = only, hilo or set(), hilo_fn
        self.names = names
        self.act_regs = {k: [re.compile(r) for r in v] for k, v in (act_split or {}).items()}
        self.inner_regs = {k: [re.compile(r) for r in v] for k, v in (inner16 or {}).items()}
        self.qk_regs = {k: [re.compile(r) for r in v] for k, v in (qk or {}).items()}      # attention blocks whose q / k carry two-term splits
        self.stage = "lat"
        self.seen = set()
        self.seen4 = set()
        self.unnamed = set()

    @staticmethod
    def _q(x, dt):
        return x if dt is None else x.to(dt).float()

    attn_exact = ()          # stages whose attention internals (q, k, v, P roundings) are kept exact: --attn-exact enc,dec

    def op(self, x, kind="conv", sub="", mod=None):
        if kind == "attn" and self.stage in self.attn_exact:
            return x
        if self.names is not None and kind != "attn":
            if mod is None:
                if kind != "lat":
                    self.unnamed.add((self.stage, kind, sub))
                return x                     # latent-sized tensors stay fp32 / are split everywhere in the product
            model, name = self.names[id(mod)]
            if any(r.search(name) for r in self.act_regs.get(model, ())):
                hi = self._q(x, self.operand)
                return hi + self._q(x - hi, self.operand)
            return self._q(x, self.operand)
        key = (self.stage, kind)
        res = x.shape[-1] if x.dim() == 4 else int(round((x.shape[-2]) ** 0.5))    # spatial side ([B,C,H,W] or [B,L,C] tokens)
        key3 = (self.stage, kind, res)
        key4 = (self.stage, kind, res, sub)
        self.seen.add(key3)
        self.seen4.add(key4)
        if self.only is not None and key not in self.only and key3 not in self.only and key4 not in self.only:
            return x
        if key in self.hilo or key3 in self.hilo or key4 in self.hilo or (self.stage, "*") in self.hilo or (self.hilo_fn and self.hilo_fn(*key4)):      # two-term split: hi + lo, both in the operand type
            hi = self._q(x, self.operand)
            return hi + self._q(x - hi, self.operand)
        return self._q(x, self.operand)

    def qk_is_split(self, attn_mod) -> bool:
        if self.names is None or id(attn_mod) not in self.names:
            return False
        model, name = self.names[id(attn_mod)]
        return any(r.search(name) for r in self.qk_regs.get(model, ()))

    def split2(self, x):
        hi = self._q(x, self.operand)
        return hi + self._q(x - hi, self.operand)

    def inn(self, x, mod=None):
        if self.names is not None:           # name-based: the PRODUCING conv's output stays 16-bit where the inner16 list says so
            model, name = self.names[id(mod)]
            return self._q(x, self.operand) if any(r.search(name) for r in self.inner_regs.get(model, ())) else x
        return self._q(x, self.inner)

    def st(self, x):
        return self._q(x, self.stream)


def resnet(S: Scheme, r, x, temb=None):
    a = S.op(F.silu(r.norm1(x)), "conv", "n1", r.conv1)
    h = r.conv1(a)
    if r.time_emb_proj is not None:
        h = h + r.time_emb_proj(F.silu(temb))[:, :, None, None]
    h = S.inn(h, r.conv1)
    b = S.op(F.silu(r.norm2(h)), "conv", "n2", r.conv2)
    sc = x if r.conv_shortcut is None else r.conv_shortcut(S.op(x, "conv", "sc", r.conv_shortcut))
    return S.st(sc + r.conv2(b))


def attn_unet(S: Scheme, at, n, ctx=None):
    c = n if ctx is None else S.op(ctx, "lin", "ctx", at.to_k)
    q, k, v = S.op(at.to_q(n), "attn"), S.op(at.to_k(c), "attn"), S.op(at.to_v(c), "attn")
    q, k, v = at._heads(q), at._heads(k), at._heads(v)
    p = S.op((torch.matmul(q, k.transpose(-1, -2)) * at.scale).softmax(dim=-1), "attn")
    o = torch.matmul(p, v).transpose(1, 2).reshape(n.shape[0], -1, at.heads * at.dim_head)
    return at.to_out[0](S.op(o, "lin", "o", at.to_out[0]))


def transformer2d(S: Scheme, t, x, ehs):
    B, Cc, H, W = x.shape
    y = S.op(t.norm(x), "lin", "gn", t.proj_in).permute(0, 2, 3, 1).reshape(B, H * W, Cc)
    y = S.st(t.proj_in(y))
    for blk in t.transformer_blocks:
        y = S.st(y + attn_unet(S, blk.attn1, S.op(blk.norm1(y), "lin", "ln1", blk.attn1.to_q)))
        y = S.st(y + attn_unet(S, blk.attn2, S.op(blk.norm2(y), "lin", "ln2", blk.attn2.to_q), ehs))
        n = S.op(blk.norm3(y), "lin", "ln3", blk.ff.net[0].proj)
        hg, gate = blk.ff.net[0].proj(n).chunk(2, dim=-1)
        y = S.st(y + blk.ff.net[2](S.op(hg * F.gelu(gate), "lin", "ffh", blk.ff.net[2])))
    y = t.proj_out(S.op(y, "lin", "y", t.proj_out)).reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    return S.st(y + x)


def unet(S: Scheme, u, sample, timestep, ehs):
    B = sample.shape[0]
    t = torch.as_tensor([timestep], dtype=torch.int64).reshape(-1).expand(B)
    emb = u.time_embedding(R.timestep_sinusoid(t, u.config.block_out_channels[0]))
    if ehs.shape[0] != B:
        ehs = ehs.expand(B, -1, -1)
    h = S.st(u.conv_in(S.op(sample, "lat", "", u.conv_in)))
    skips = [h]
    for blk in u.down_blocks:
        for j, r in enumerate(blk.resnets):
            h = resnet(S, r, h, emb)
            if blk.attentions is not None:
                h = transformer2d(S, blk.attentions[j], h, ehs)
            skips.append(h)
        if blk.downsamplers is not None:
            h = S.st(blk.downsamplers[0](S.op(h, "conv", "samp", blk.downsamplers[0].conv)))
            skips.append(h)
    m = u.mid_block
    h = resnet(S, m.resnets[0], h, emb)
    h = transformer2d(S, m.attentions[0], h, ehs)
    h = resnet(S, m.resnets[1], h, emb)
    for blk in u.up_blocks:
        for j, r in enumerate(blk.resnets):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(S, r, h, emb)
            if blk.attentions is not None:
                h = transformer2d(S, blk.attentions[j], h, ehs)
        if blk.upsamplers is not None:
            h = S.st(blk.upsamplers[0](S.op(h, "conv", "samp", blk.upsamplers[0].conv)))
    return S.op(u.conv_out(S.op(F.silu(u.conv_norm_out(h)), "conv", "", u.conv_out)), "lat")


def vae_attn(S: Scheme, at, x):
    B, Cc, H, W = x.shape
    g = S.op(at.group_norm(x.view(B, Cc, H * W)), "lin", "", at.to_q).transpose(1, 2)
    # --attn-exact enc:qk / dec:qk / enc:pv / dec:pv: only the q, k (or P, v) roundings of that stage's attention stay exact
    qk_exact, pv_exact = (S.stage + ":qk") in S.attn_exact, (S.stage + ":pv") in S.attn_exact
    q, k = at.to_q(g), at.to_k(g)
    if not qk_exact:
        q, k = (S.split2(q), S.split2(k)) if S.qk_is_split(at) else (S.op(q, "attn"), S.op(k, "attn"))
    v = at.to_v(g) if pv_exact else S.op(at.to_v(g), "attn")
    p = (torch.matmul(q, k.transpose(-1, -2)) * at.scale).softmax(dim=-1)
    if not pv_exact:
        p = S.op(p, "attn")
    o = at.to_out[0](S.op(torch.matmul(p, v), "lin", "", at.to_out[0]))
    return S.st(o.transpose(-1, -2).reshape(B, Cc, H, W) + x)


def vae_mid(S, m, h):
    h = resnet(S, m.resnets[0], h)
    h = vae_attn(S, m.attentions[0], h)
    return resnet(S, m.resnets[1], h)


def encoder(S: Scheme, e, x):
    h = S.st(e.conv_in(S.op(x, "lat", "", e.conv_in)))
    for b in e.down_blocks:
        for r in b.resnets:
            h = resnet(S, r, h)
        if b.downsamplers is not None:
            h = S.st(b.downsamplers[0](S.op(h, "conv", "samp", b.downsamplers[0].conv)))
    h = vae_mid(S, e.mid_block, h)
    return e.conv_out(S.op(F.silu(e.conv_norm_out(h)), "conv", "", e.conv_out))


def decoder(S: Scheme, d, z):
    h = S.st(d.conv_in(S.op(z, "lat", "", d.conv_in)))
    h = vae_mid(S, d.mid_block, h)
    for b in d.up_blocks:
        for r in b.resnets:
            h = resnet(S, r, h)
        if b.upsamplers is not None:
            h = S.st(b.upsamplers[0](S.op(h, "conv", "samp", b.upsamplers[0].conv)))
    return S.op(d.conv_out(S.op(F.silu(d.conv_norm_out(h)), "conv", "", d.conv_out)), "lat")


def omgsr_s(S: Scheme, vae, u, alpha_t, x, ehs, eps, tile, overlap):
    sf = vae.config.scaling_factor
    S.stage = "enc"
    m = vae.quant_conv(S.op(encoder(S, vae.encoder, x), "lat", "", vae.quant_conv))
    S.stage = "lat"
    mean, logvar = m.chunk(2, dim=1)
    z = S.op((mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * eps) * sf, "lat")
    S.stage = "unet"
    _, c, h, w = z.shape
    if h * w <= tile * tile:
        pred = unet(S, u, z, 273, ehs)
    else:
        ts, _, _, offs = tile_offsets(h, w, tile, overlap)
        preds = [unet(S, u, z[:, :, oy:oy + ts, ox:ox + ts], 273, ehs) for (oy, ox) in offs]
        pred = S.op(stitch(z.shape, preds, offs, ts, c), "lat")
    S.stage = "lat"
    z0 = S.op((z - (1 - alpha_t).sqrt() * pred) / alpha_t.sqrt() / sf, "lat")
    S.stage = "dec"
    return decoder(S, vae.decoder, vae.post_quant_conv(S.op(z0, "lat", "", vae.post_quant_conv))).clamp(-1, 1)


def policy_r2_first(stage, kind, res, sub):
    """The first accurate-tier policy: every UNet level but the 8 x 8 one, the encoder's 512-px level, latent-sized tensors."""
    if kind == "lat":
        return True
    if stage == "unet":
        return res >= 16 and kind in ("conv", "lin")
    return stage == "enc" and kind == "conv" and res == 512


def policy_full_signal(stage, kind, res, sub):
    """Split where the WHOLE signal passes through one operand rounding (1x1 shortcuts, up / down-sampling convs of the encoder
    and the UNet, proj_in / proj_out, conv_in / conv_out, latents) plus the UNet's 64 x 64 level."""
    if kind == "lat" or sub == "sc" or (kind == "conv" and sub == ""):
        return True
    if sub == "samp":
        return stage in ("enc", "unet")
    if stage == "unet":
        return sub in ("gn", "y") or res == 64
    return False


POLICIES = {"r2_first": policy_r2_first, "full_signal": policy_full_signal}


# ---- name-based policies (the product's own lists: omgsr_amd/precision.py) with the WEIGHT side emulated too -----------------
def names_of(vae, u):
    out = {id(m): ("vae", n) for n, m in vae.named_modules()}
    out.update({id(m): ("unet", n) for n, m in u.named_modules()})
    return out


@torch.no_grad()
def weights_as_packed(model, w_split, dt=torch.float16):
    """A copy of `model` holding the weights the MFMAs see: every Conv2d / Linear weight rounded to `dt`, or, where a w_split
    pattern matches, its two-term split w_hi + w_lo. The time embedding / time_emb_proj linears are folded in fp32 by the product
    (constant at fixed t*) and stay exact."""
    m2 = type(model)(**dict(model.config)).eval()        # (oracle Config objects do not deep-copy)
    m2.load_state_dict(model.state_dict())
    regs = [re.compile(r) for r in w_split]
    for name, mod in m2.named_modules():
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)) and "time_emb" not in name:
            w = mod.weight.data
            hi = w.to(dt).float()
            mod.weight.data = hi + (w - hi).to(dt).float() if any(r.search(name) for r in regs) else hi
    return m2


def named_policies():
    from omgsr_amd import precision as Pn
    base = dict(act=dict(vae=Pn.VAE_DEFAULT, unet=Pn.UNET_DEFAULT), inner16=dict(vae=Pn.VAE_INNER16), w=dict(vae=[], unet=[]))
    out = {"r2": base,
           "r2_wsame": dict(base, w=dict(vae=Pn.VAE_DEFAULT, unet=Pn.UNET_DEFAULT)),
           "all_w": dict(base, w=dict(vae=[r"."], unet=[r"."])),
           "all_aw": dict(act=dict(vae=[r"."], unet=[r"."]), inner16={}, w=dict(vae=[r"."], unet=[r"."]))}
    for k in ("VAE_ACT", "UNET_ACT", "VAE_W", "UNET_W"):
        if not hasattr(Pn, k):
            return out
    out["shipped"] = dict(act=dict(vae=Pn.VAE_ACT, unet=Pn.UNET_ACT), inner16=dict(vae=Pn.VAE_INNER16), w=dict(vae=Pn.VAE_W, unet=Pn.UNET_W),
                          qk=dict(vae=getattr(Pn, "VAE_QK_SPLIT", [])))
    return out

SCHEMES = {
    "bf16_all": (torch.bfloat16,) * 3,
    "f16_all": (torch.float16,) * 3,
    "bf16_stream32": (torch.bfloat16, torch.bfloat16, None),
    "f16_stream32": (torch.float16, torch.float16, None),
    "bf16_inner32": (torch.bfloat16, None, None),
    "f16_inner32": (torch.float16, None, None),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--schemes", default="f16_all,f16_stream32,f16_inner32,bf16_inner32")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--budget-stage", default="unet")
    ap.add_argument("--budget", default="", help="f16 | bf16: per (stage, kind) contribution of the operand roundings")
    ap.add_argument("--named", default="", help="comma list of name-based policies (named_policies(), or a JSON file of them): activation "
                                                "AND weight roundings emulated")
    ap.add_argument("--fp32-weights", action="store_true", help="seeded weights with full fp32 mantissas (not pre-rounded to bf16 values)")
    ap.add_argument("--wseed", type=int, default=0)
    ap.add_argument("--xseed", type=int, default=1234)
    ap.add_argument("--test-draw", default="", help="W,X: the weights / input / prompt / noise of weight draw W, input draw X of "
                                                    "tests/test_fullsize_parity_gpu.py::test_accurate_tier_full_mantissa_weights_over_seeds")
    ap.add_argument("--attn-exact", default="", help="comma list of stages (enc, unet, dec) whose attention internals stay exact")
    a = ap.parse_args()
    Scheme.attn_exact = tuple(v for v in a.attn_exact.split(",") if v)
    torch.set_num_threads(a.threads)
    if a.test_draw:
        w_, x_ = (int(v) for v in a.test_draw.split(","))
        vae = seeded_init_(R.AutoencoderKL(), 1101 + 17 * w_, rounded=False).eval()
        u = seeded_init_(R.UNet2DConditionModel(), 2202 + 17 * w_, rounded=False).eval()
        g_ = torch.Generator().manual_seed(5000 + 10 * w_ + x_)
        x = synthetic_lq(1, 512, 512, seed=777 + 10 * w_ + x_)
        ehs = torch.randn(1, 77, 1024, generator=g_)
        eps = torch.randn(1, 4, 64, 64, generator=g_)
    else:
        vae = seeded_init_(R.AutoencoderKL(), 101 + a.wseed, rounded=not a.fp32_weights).eval()
        u = seeded_init_(R.UNet2DConditionModel(), 202 + a.wseed, rounded=not a.fp32_weights).eval()
        x = synthetic_lq(1, a.side, a.side, seed=a.xseed)
        eps = torch.randn(1, 4, a.side // 8, a.side // 8, generator=torch.Generator().manual_seed(99))
        ehs = torch.randn(1, 77, 1024, generator=torch.Generator().manual_seed(4321)).to(torch.bfloat16).float()
    alpha_t = R.DDPMScheduler().alphas_cumprod[273]
    with torch.no_grad():
        t0 = time.time()
        vae.posterior_noise = eps
        ref = OmgsrSRef(vae, u, alpha_t, 273)(x, ehs, 64, 32)
        print(f"oracle: {time.time() - t0:.1f} s, rms {ref.pow(2).mean().sqrt():.3f}", flush=True)
        same = omgsr_s(Scheme(None, None, None), vae, u, alpha_t, x, ehs, eps, 64, 32)
        print(f"emulator with no rounding vs oracle: rel-L2 {rel_l2(same, ref):.2e}", flush=True)
        if a.named:
            import json
            pols = named_policies()
            for name in a.named.split(","):
                if os.path.isfile(name):
                    pols.update(json.load(open(name)))
            for name in a.named.split(","):
                if os.path.isfile(name):
                    continue
                pol = pols[name]
                v2, u2 = weights_as_packed(vae, pol["w"]["vae"]), weights_as_packed(u, pol["w"]["unet"])
                v2.posterior_noise = eps
                S = Scheme(torch.float16, None, None, names=names_of(v2, u2), act_split=pol["act"], inner16=pol.get("inner16"), qk=pol.get("qk"))
                got = omgsr_s(S, v2, u2, alpha_t, x, ehs, eps, 64, 32)
                e = rel_l2(got, ref)
                print(f"{name:16s} rel-L2 {e:.3e}  var {e * e * 1e8:.1f}  PSNR {psnr(got, ref):.1f} dB  unnamed operands: {sorted(S.unnamed)}", flush=True)
                del v2, u2
            return
        if a.budget.endswith("_subs"):
            dt = torch.float16 if a.budget.startswith("f16") else torch.bfloat16
            probe = Scheme(dt, None, None)
            omgsr_s(probe, vae, u, alpha_t, x, ehs, eps, 64, 32)
            want = (lambda k: k[0] == "unet" and k[2] in (64, 32)) if a.budget_stage == "unet" else (lambda k: k[0] == a.budget_stage)
            for key4 in sorted(k for k in probe.seen4 if want(k) and k[1] in ("conv", "lin")):
                got = omgsr_s(Scheme(dt, None, None, only={key4}), vae, u, alpha_t, x, ehs, eps, 64, 32)
                e = rel_l2(got, ref)
                print(f"only {key4} operand roundings: rel-L2 {e:.3e}  var {e * e * 1e8:.1f}", flush=True)
            return
        if a.budget.endswith("_levels"):
            dt = torch.float16 if a.budget.startswith("f16") else torch.bfloat16
            probe = Scheme(dt, None, None)
            omgsr_s(probe, vae, u, alpha_t, x, ehs, eps, 64, 32)
            tot = 0.0
            for key3 in sorted(k for k in probe.seen if k[1] in ("conv", "lin")):
                got = omgsr_s(Scheme(dt, None, None, only={key3}), vae, u, alpha_t, x, ehs, eps, 64, 32)
                e = rel_l2(got, ref)
                tot += e * e
                print(f"only {key3} operand roundings: rel-L2 {e:.3e}  var {e * e * 1e8:.1f}", flush=True)
            print(f"quadrature sum {tot ** 0.5:.3e}")
            return
        if a.budget:
            dt = torch.float16 if a.budget == "f16" else torch.bfloat16
            tot = 0.0
            for stage in ("enc", "unet", "dec", "lat"):
                for kind in ("conv", "lin", "attn", "lat"):
                    got = omgsr_s(Scheme(dt, None, None, only={(stage, kind)}), vae, u, alpha_t, x, ehs, eps, 64, 32)
                    e = rel_l2(got, ref)
                    tot += e * e
                    if e > 0:
                        print(f"only ({stage:4s},{kind:4s}) operand roundings: rel-L2 {e:.3e}", flush=True)
            print(f"quadrature sum {tot ** 0.5:.3e}")
            return
        for name in a.schemes.split(","):
            if name.startswith("policy:"):
                got = omgsr_s(Scheme(torch.float16, None, None, hilo_fn=POLICIES[name[7:]]), vae, u, alpha_t, x, ehs, eps, 64, 32)
                print(f"{name:16s} rel-L2 {rel_l2(got, ref):.3e}  PSNR {psnr(got, ref):.1f} dB", flush=True)
                continue
            got = omgsr_s(Scheme(*SCHEMES[name]), vae, u, alpha_t, x, ehs, eps, 64, 32)
            print(f"{name:16s} rel-L2 {rel_l2(got, ref):.3e}  PSNR {psnr(got, ref):.1f} dB", flush=True)


if __name__ == "__main__":
    main()

"""SURVEY §8(f) f3 — the constant cache, serialised.

OMGSR runs its denoiser at ONE timestep with ONE prompt (infer/infer_omgsr_s.py:19-45 encodes the prompt once with the CLIP
text encoder and frees it; infer/infer_omgsr_f.py:36-48 does the same with T5-XXL + CLIP-L through FluxPipeline). Everything that
depends only on (weights, t*, prompt) is therefore a constant of the deployment:

  OMGSR-S   prompt_embeds [1,77,1024]; every ResnetBlock's conv1 bias with time_emb_proj(silu(time_embedding(sinusoid(t*)))) folded
            in; K and V^T of every cross-attention layer
  OMGSR-F   prompt_embeds [1,512,4096], pooled [1,768], text / image ids; the context embedding; the RoPE tables; every
            AdaLN-Zero modulation vector (3.28 B parameters that only ever see the 1-token constant temb)

This module writes them to ONE safetensors file next to the weights and loads them back into the modules' caches, so an
inference host needs neither `transformers`, the 4.7 B-parameter T5 nor a warm-up forward: `load_s(pipe, path)` returns the
prompt tensor to pass to `pipe(...)`.

Wire format (version 2): a safetensors container; `__metadata__` holds
  format = "omgsr-constants", version, family ("S" | "F"), tier (bf16 | fp16 | fp32: ops.compute_dtype_name()), abi (C ABI version),
  mid_timestep, (F: guidance_scale, t_curr), checksum.<module> = bit-exact checksum of the weights the constants were folded from,
  policy.<module> = fingerprint of the accurate tier's precision policy they were folded under ("none" in the fast tiers)
and the tensors are named `<model>.<kind>.<qualified module name>[.<field>]` (listed by `describe(path)`). A file whose tier,
timestep, weight checksum or precision policy does not match the pipeline it is loaded into is refused.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import _lib, ops
from .dist import module_checksum

FORMAT, VERSION = "omgsr-constants", "2"


class ConstantsMismatch(RuntimeError):
    pass


def _checksum(m: torch.nn.Module) -> str:
    c = module_checksum(m).cpu().tolist()
    return f"{c[0]:x}:{c[1]:x}"


def _meta(family: str, pipe, extra: Dict[str, str], modules: Dict[str, torch.nn.Module]) -> Dict[str, str]:
    meta = {"format": FORMAT, "version": VERSION, "family": family, "tier": ops.compute_dtype_name(), "abi": str(_lib.ABI_VERSION),
            "mid_timestep": str(int(pipe.mid_timestep))}
    meta.update(extra)
    from .precision import policy_fingerprint
    for name, m in modules.items():
        meta[f"checksum.{name}"] = _checksum(m)
        # the folded K / V^T / context tensors are GEMM outputs: they depend on which layers carry split operands / weights
        meta[f"policy.{name}"] = policy_fingerprint(m) if ops.precise() else "none"
    return meta


def write(path: str, tensors: Dict[str, torch.Tensor], meta: Dict[str, str]) -> None:
    from safetensors.torch import save_file
    save_file({k: v.detach().contiguous().cpu() for k, v in tensors.items()}, path, metadata=meta)


def read(path: str, device) -> Tuple[Dict[str, torch.Tensor], Dict[str, str]]:
    from safetensors import safe_open
    out = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        for k in f.keys():
            out[k] = f.get_tensor(k).to(device)
    if meta.get("format") != FORMAT:
        raise ConstantsMismatch(f"{path}: not an {FORMAT} file")
    if meta.get("version") != VERSION:
        raise ConstantsMismatch(f"{path}: wire format version {meta.get('version')}, this build reads {VERSION}")
    return out, meta


def describe(path: str) -> Dict[str, object]:
    t, m = read(path, "cpu")
    return {"metadata": m, "tensors": {k: (tuple(v.shape), str(v.dtype)) for k, v in sorted(t.items())}}


def _validate(meta: Dict[str, str], family: str, pipe, modules: Dict[str, torch.nn.Module], extra: Dict[str, str]) -> None:
    want = _meta(family, pipe, extra, modules)
    for k, v in want.items():
        if meta.get(k) != v:
            raise ConstantsMismatch(f"constant cache does not belong to this pipeline: {k} is {meta.get(k)!r} in the file, {v!r} here")


# ---- OMGSR-S ---------------------------------------------------------------------------------------------------------
def _s_cross_attention(unet):
    from .diffusers_api.unet_2d_condition import Attention
    return [(n, m) for n, m in unet.named_modules() if isinstance(m, Attention) and m.is_cross]


@torch.no_grad()
def collect_s(pipe, prompt_embeds: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Fold the constants of (pipe.unet, pipe.mid_timestep, prompt_embeds). Runs the K / V^T projection GEMMs: needs the GPU."""
    unet = pipe.unet
    names = {id(m): n for n, m in unet.named_modules()}
    out = {"prompt_embeds": prompt_embeds}
    fb = unet._folded_biases(pipe.mid_timestep)
    for r in unet._resnets():
        out[f"unet.fold.{names[id(r)]}.conv1_bias"] = fb[id(r)]
    for n, at in _s_cross_attention(unet):
        kk, vt, _ = at.context(prompt_embeds)
        out[f"unet.ctx.{n}.k"], out[f"unet.ctx.{n}.vt"] = kk, vt
    return out


def export_s(pipe, prompt_embeds: torch.Tensor, path: str) -> None:
    write(path, collect_s(pipe, prompt_embeds), _meta("S", pipe, {}, {"unet": pipe.unet}))


def load_s(pipe, path: str) -> torch.Tensor:
    """Install a constant cache into pipe.unet; returns the prompt_embeds tensor the caches are keyed on (pass THAT tensor)."""
    dev = next(pipe.unet.parameters()).device
    t, meta = read(path, dev)
    _validate(meta, "S", pipe, {"unet": pipe.unet}, {})
    unet = pipe.unet
    names = {id(m): n for n, m in unet.named_modules()}
    prompt = t["prompt_embeds"]
    unet._temb_cache = {unet._fold_key(pipe.mid_timestep): {id(r): t[f"unet.fold.{names[id(r)]}.conv1_bias"].float().contiguous()
                                                             for r in unet._resnets()}}
    for n, at in _s_cross_attention(unet):
        at._ctx_cache.prime((prompt,), at._ctx_key(), (t[f"unet.ctx.{n}.k"].contiguous(), t[f"unet.ctx.{n}.vt"].contiguous(), prompt.shape[1]))
    return prompt


# ---- OMGSR-F ---------------------------------------------------------------------------------------------------------
def _f_extra(pipe) -> Dict[str, str]:
    return {"guidance_scale": repr(float(pipe.guidance_scale)), "t_curr": repr(float(pipe.t_curr))}


def _f_call_args(pipe, dev):
    timestep = torch.tensor([pipe.t_curr], device=dev)
    guidance = torch.full((1,), pipe.guidance_scale, device=dev, dtype=torch.float32)
    return timestep, guidance


@torch.no_grad()
def collect_f(pipe, prompt_embeds, pooled, text_ids, image_ids) -> Dict[str, torch.Tensor]:
    flux = pipe.flux_transformer
    dev = next(flux.parameters()).device
    timestep, guidance = _f_call_args(pipe, dev)
    mod = flux._modulation(timestep, guidance, pooled)
    cos, sin = flux._rope(text_ids, image_ids)
    out = {"prompt_embeds": prompt_embeds, "pooled": pooled, "text_ids": text_ids, "image_ids": image_ids,
           "flux.ctx": flux._context(prompt_embeds), "flux.rope.cos": cos, "flux.rope.sin": sin,
           "flux.mod.out.a": mod["out"]["a"], "flux.mod.out.b": mod["out"]["b"]}
    for i, m in enumerate(mod["double"]):
        for stream in ("img", "ctx"):
            for k, v in m[stream].items():
                out[f"flux.mod.double.{i}.{stream}.{k}"] = v
    for i, m in enumerate(mod["single"]):
        for k, v in m.items():
            out[f"flux.mod.single.{i}.{k}"] = v
    return out


def export_f(pipe, prompt_embeds, pooled, text_ids, image_ids, path: str) -> None:
    write(path, collect_f(pipe, prompt_embeds, pooled, text_ids, image_ids), _meta("F", pipe, _f_extra(pipe), {"flux": pipe.flux_transformer}))


def load_f(pipe, path: str):
    """Install a constant cache into pipe.flux_transformer; returns (prompt_embeds, pooled, text_ids, image_ids) to call it with."""
    flux = pipe.flux_transformer
    dev = next(flux.parameters()).device
    t, meta = read(path, dev)
    _validate(meta, "F", pipe, {"flux": flux}, _f_extra(pipe))
    prompt, pooled, tids, iids = t["prompt_embeds"], t["pooled"], t["text_ids"], t["image_ids"]
    timestep, guidance = _f_call_args(pipe, dev)
    nd, ns = len(flux.transformer_blocks), len(flux.single_transformer_blocks)
    mod = {"double": [{s: {k: t[f"flux.mod.double.{i}.{s}.{k}"].contiguous() for k in ("a1", "b1", "g1", "a2", "b2", "g2")} for s in ("img", "ctx")}
                      for i in range(nd)],
           "single": [{k: t[f"flux.mod.single.{i}.{k}"].contiguous() for k in ("a", "b", "g")} for i in range(ns)],
           "out": {"a": t["flux.mod.out.a"].contiguous(), "b": t["flux.mod.out.b"].contiguous()}}
    flux._mod_cache.prime((pooled,), flux._mod_key(timestep, guidance), mod)
    flux._rope_cache.prime((tids, iids), (), (t["flux.rope.cos"].contiguous(), t["flux.rope.sin"].contiguous()))
    flux._ctx_cache.prime((prompt,), flux._ctx_key(), t["flux.ctx"].contiguous())
    return prompt, pooled, tids, iids

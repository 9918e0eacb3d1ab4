"""ModelMixin / config plumbing shared by the diffusers-shaped modules: config dict with attribute
access, `from_config`, `from_pretrained(path, subfolder=...)` over the HF directory layout
(`<root>/<subfolder>/config.json` + `diffusion_pytorch_model*.safetensors`),
`.dtype` / `.device` properties — the surface infer/omgsr_{s,f}_infer_model.py touch.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn


class ConfigDict(dict):
    """dict with attribute access (stand-in for diffusers' FrozenDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ModelMixin(nn.Module):
    config_name = "config.json"
    default_config: dict = {}

    @property
    def dtype(self) -> torch.dtype:
        for p in self.parameters():
            return p.dtype
        return torch.float32

    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    @classmethod
    def from_config(cls, config: dict | None = None, **kwargs):
        cfg = {k: v for k, v in (config or {}).items() if not k.startswith("_")}
        cfg.update(kwargs)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: str | None = None, torch_dtype: torch.dtype | None = None, **_):
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, cls.config_name)) as f:
            cfg = json.load(f)
        model = cls.from_config(cfg)
        sd = convert_legacy_keys(load_safetensors_dir(root), model)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise RuntimeError(f"{cls.__name__}.from_pretrained({root}): missing keys {missing[:5]}..., unexpected {unexpected[:5]}...")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    def save_pretrained(self, path: str, subfolder: str | None = None, max_shard_size: int | None = None):
        """HF layout: one `diffusion_pytorch_model.safetensors`, or (max_shard_size bytes given and exceeded) numbered shards +
        `diffusion_pytorch_model.safetensors.index.json` with the `weight_map` diffusers writes (FLUX.1-dev ships 3 shards)."""
        from safetensors.torch import save_file
        root = os.path.join(path, subfolder) if subfolder else path
        os.makedirs(root, exist_ok=True)
        with open(os.path.join(root, self.config_name), "w") as f:
            json.dump({"_class_name": type(self).__name__, **dict(self.config)}, f, indent=1)
        sd = {k: v.contiguous() for k, v in self.state_dict().items()}
        total = sum(v.numel() * v.element_size() for v in sd.values())
        if not max_shard_size or total <= max_shard_size:
            save_file(sd, os.path.join(root, "diffusion_pytorch_model.safetensors"))
            return
        shards, cur, size = [], {}, 0
        for k, v in sd.items():
            n = v.numel() * v.element_size()
            if cur and size + n > max_shard_size:
                shards.append(cur)
                cur, size = {}, 0
            cur[k] = v
            size += n
        shards.append(cur)
        weight_map = {}
        for i, sh in enumerate(shards):
            name = f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(root, name))
            weight_map.update({k: name for k in sh})
        with open(os.path.join(root, "diffusion_pytorch_model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)


_LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def convert_legacy_keys(sd: dict, model: nn.Module) -> dict:
    """Checkpoint key names older than the module tree, converted the way diffusers does at load time
    (`_convert_deprecated_attention_blocks`): the SD2.1-base VAE stores its mid-block attention as
    `mid_block.attentions.0.{query,key,value,proj_attn}.{weight,bias}`; some VAEs keep those projections as 1x1
    convolution weights [C, C, 1, 1]. Keys the model already knows are left alone."""
    want = model.state_dict()
    out = {}
    for k, v in sd.items():
        nk = k
        if k not in want:
            parts = k.split(".")
            for i, p in enumerate(parts):
                if p in _LEGACY_ATTN and i > 0 and parts[i - 1].isdigit() and "attentions" in parts[:i]:
                    cand = ".".join(parts[:i] + [_LEGACY_ATTN[p]] + parts[i + 1:])
                    if cand in want:
                        nk = cand
                    break
        if nk in want and v.dim() == 4 and want[nk].dim() == 2 and v.shape[2:] == (1, 1):
            v = v[:, :, 0, 0]
        out[nk] = v
    return out


def load_safetensors_dir(root: str) -> dict:
    """Single-file or sharded (`*.safetensors.index.json`) diffusers checkpoint -> state dict."""
    from safetensors.torch import load_file
    single = os.path.join(root, "diffusion_pytorch_model.safetensors")
    if os.path.isfile(single):
        return load_file(single)
    index = os.path.join(root, "diffusion_pytorch_model.safetensors.index.json")
    if os.path.isfile(index):
        with open(index) as f:
            shards = sorted(set(json.load(f)["weight_map"].values()))
        sd = {}
        for s in shards:
            sd.update(load_file(os.path.join(root, s)))
        return sd
    raise FileNotFoundError(f"no diffusion_pytorch_model*.safetensors under {root}")

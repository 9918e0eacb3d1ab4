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
        sd = load_safetensors_dir(root)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise RuntimeError(f"{cls.__name__}.from_pretrained({root}): missing keys {missing[:5]}..., unexpected {unexpected[:5]}...")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    def save_pretrained(self, path: str, subfolder: str | None = None):
        from safetensors.torch import save_file
        root = os.path.join(path, subfolder) if subfolder else path
        os.makedirs(root, exist_ok=True)
        with open(os.path.join(root, self.config_name), "w") as f:
            json.dump({"_class_name": type(self).__name__, **dict(self.config)}, f, indent=1)
        save_file({k: v.contiguous() for k, v in self.state_dict().items()}, os.path.join(root, "diffusion_pytorch_model.safetensors"))


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

"""PeftModel stand-in: exactly the protocol infer/omgsr_s_infer_model.py:16-23 and
infer/omgsr_f_infer_model.py:115-118 exercise —
`PeftModel.from_pretrained(module, adapter_dir[, is_trainable=False])` returns a wrapper that forwards
calls/attributes to the base module; `.merge_and_unload()` folds W += (lora_alpha / r) * B @ A into the
base weights IN PLACE (the reference discards the return value, so in-place is what matters).
After the merge there is no LoRA math at inference time.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn

from .. import ops


class PeftModel(nn.Module):
    def __init__(self, base: nn.Module, adapter_sd: dict, r: int, lora_alpha: float):
        super().__init__()
        self.base_model = base
        self._adapter_sd, self._r, self._alpha = adapter_sd, r, lora_alpha
        self._merged = False

    @classmethod
    def from_pretrained(cls, model: nn.Module, model_id: str, is_trainable: bool = False, **_):
        from safetensors.torch import load_file
        with open(os.path.join(model_id, "adapter_config.json")) as f:
            cfg = json.load(f)
        sd = load_file(os.path.join(model_id, "adapter_model.safetensors"))
        return cls(model, sd, int(cfg["r"]), float(cfg.get("lora_alpha", cfg["r"])))

    def forward(self, *a, **kw):
        return self.base_model(*a, **kw)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("base_model"), name)

    @staticmethod
    def _target_name(key: str) -> str:
        name = key[: -len(".lora_A.weight")]
        for prefix in ("base_model.model.", "base_model."):
            if name.startswith(prefix):
                return name[len(prefix):]
        return name

    @torch.no_grad()
    def merge_and_unload(self):
        if self._merged:
            return self.base_model
        scale = self._alpha / self._r
        mods = dict(self.base_model.named_modules())
        for key, A in self._adapter_sd.items():
            if not key.endswith(".lora_A.weight"):
                continue
            Bm = self._adapter_sd[key.replace(".lora_A.", ".lora_B.")]
            tgt = mods[self._target_name(key)]
            w = tgt.weight
            A32, B32 = A.to(w.device, torch.float32), Bm.to(w.device, torch.float32)
            # linear: B [out, r] @ A [r, in]; conv: B [out, r, 1, 1], A [r, in, k, k] - one rank-r product either way, in the library's
            # fixed-order fp32 kernel when the weights already live on the GPU (no vendor BLAS in the product process)
            delta = ops.linear_f32(B32.reshape(B32.shape[0], -1), A32.reshape(A32.shape[0], -1).t()).reshape(w.shape)
            w.add_((scale * delta).to(w.dtype))
        self._merged = True
        return self.base_model

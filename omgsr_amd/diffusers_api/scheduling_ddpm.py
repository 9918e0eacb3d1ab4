"""DDPMScheduler: only what infer/omgsr_s_infer_model.py:13-14 uses — `alphas_cumprod[t]`
(scaled-linear betas, SD2.1-base scheduler_config.json)."""
from __future__ import annotations

import json
import os

import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "epsilon", **_):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, prediction_type=prediction_type)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: str | None = None, **_):
        root = os.path.join(path, subfolder) if subfolder else path
        cfg_path = os.path.join(root, "scheduler_config.json")
        cfg = {}
        if os.path.isfile(cfg_path):
            with open(cfg_path) as f:
                cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        keep = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type")
        return cls(**{k: v for k, v in cfg.items() if k in keep})

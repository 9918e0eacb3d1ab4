"""ORACLE — test infrastructure only (never imported by omgsr_amd/).

FluxTransformer2DModel.forward of oracle/diffusers_ref.py (diffusers 0.34.0 `transformer_flux.py`; reference call
site infer/omgsr_f_infer_model.py:191-200) with ONE block's weights resident at a time, so that the fp32 CPU oracle of the
full-depth FLUX.1-dev DiT (19 double + 38 single blocks, 11.9 B parameters = 48 GB in fp32) runs in ~1.5 GB of host memory:
the weights of each block are fetched from a caller-supplied source (`fetch(prefix)` -> {key: fp32 CPU tensor}), loaded into a
freshly built oracle block, used once and dropped. The arithmetic is exactly that of `diffusers_ref.FluxTransformer2DModel`
(tests/test_oracle_golden.py::test_streamed_flux_equals_resident checks bit equality on a small configuration).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Dict

import torch

from . import diffusers_ref as R


def _build(ctor, sd: Dict[str, torch.Tensor]):
    with torch.device("meta"):
        m = ctor()
    m.load_state_dict(sd, assign=True)
    return m.eval()


def module_fetcher(model: torch.nn.Module) -> Callable[[str], Dict[str, torch.Tensor]]:
    """fetch() over any module whose state-dict keys are diffusers' (an oracle model, or a model living on another device)."""
    def fetch(prefix: str) -> Dict[str, torch.Tensor]:
        sub = model.get_submodule(prefix) if prefix else model
        return {k: v.detach().to(device="cpu", dtype=torch.float32) for k, v in sub.state_dict().items()}
    return fetch


class StreamedFlux:
    def __init__(self, fetch: Callable[[str], Dict[str, torch.Tensor]], **cfg):
        self.fetch = fetch
        self.config = R.Config({**R.FLUX_DEV_CONFIG, **cfg})
        self.dtype = torch.float32
        self.trace = None          # optional list: receives (hidden_states, velocity) of every call

    @torch.no_grad()
    def __call__(self, hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids,
                 return_dict: bool = True):
        c_ = self.config
        dim = c_.num_attention_heads * c_.attention_head_dim
        H, Dh = c_.num_attention_heads, c_.attention_head_dim
        lin = lambda i, o: (lambda: torch.nn.Linear(i, o))          # noqa: E731
        h = _build(lin(c_.in_channels, dim), self.fetch("x_embedder"))(hidden_states)
        B = h.shape[0]
        timestep = timestep.to(h.dtype) * 1000
        guidance = guidance.to(h.dtype) * 1000 if guidance is not None else None
        tte = _build(lambda: R._TimeTextEmbed(dim, c_.pooled_projection_dim, c_.guidance_embeds), self.fetch("time_text_embed"))
        temb = tte(timestep, guidance, pooled_projections)
        del tte
        if temb.shape[0] != B:
            temb = temb.expand(B, -1)
        c = _build(lin(c_.joint_attention_dim, dim), self.fetch("context_embedder"))(encoder_hidden_states)
        if c.shape[0] != B:
            c = c.expand(B, -1, -1)
        rope = R.flux_pos_embed(torch.cat([txt_ids, img_ids], dim=0), c_.axes_dims_rope)
        for i in range(c_.num_layers):
            blk = _build(lambda: R.FluxTransformerBlock(dim, H, Dh), self.fetch(f"transformer_blocks.{i}"))
            c, h = blk(h, c, temb, rope)
            del blk
        n_txt = c.shape[1]
        x = torch.cat([c, h], dim=1)
        for i in range(c_.num_single_layers):
            blk = _build(lambda: R.FluxSingleTransformerBlock(dim, H, Dh), self.fetch(f"single_transformer_blocks.{i}"))
            x = blk(x, temb, rope)
            del blk
        h = x[:, n_txt:]
        no = _build(lambda: R.AdaLayerNormContinuous(dim), self.fetch("norm_out"))
        po = _build(lin(dim, c_.patch_size * c_.patch_size * (c_.out_channels or c_.in_channels)), self.fetch("proj_out"))
        out = po(no(h, temb))
        if self.trace is not None:
            self.trace.append((hidden_states.clone(), out.clone()))
        return SimpleNamespace(sample=out) if return_dict else (out,)

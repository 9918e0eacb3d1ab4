"""SURVEY §8(f) f3 — the constant-cache wire format (omgsr_amd/constants.py) on CPU: container round trip, metadata, cache
priming, and the refusal of a file that does not belong to the pipeline (other timestep / tier / weights)."""
from types import SimpleNamespace

import pytest
import torch

SMALL_UNET = dict(block_out_channels=[32, 64, 64, 64], attention_head_dim=[1, 2, 2, 2], cross_attention_dim=64)


def _pipe(seed=3):
    from omgsr_amd.diffusers_api import UNet2DConditionModel
    from omgsr_amd.testing import seeded_init_
    return SimpleNamespace(unet=seeded_init_(UNet2DConditionModel(**SMALL_UNET), seed).eval(), mid_timestep=273)


def _fake_s_tensors(pipe, prompt):
    from omgsr_amd import constants as K
    names = {id(m): n for n, m in pipe.unet.named_modules()}
    t = {"prompt_embeds": prompt}
    for i, r in enumerate(pipe.unet._resnets()):
        t[f"unet.fold.{names[id(r)]}.conv1_bias"] = torch.full((r.conv1.out_channels,), float(i))
    for n, at in K._s_cross_attention(pipe.unet):
        t[f"unet.ctx.{n}.k"] = torch.randn(1, 77, at.inner).to(torch.bfloat16)
        t[f"unet.ctx.{n}.vt"] = torch.randn(1, at.inner, 80).to(torch.bfloat16)
    return t


def test_round_trip_primes_the_caches(tmp_path):
    from omgsr_amd import constants as K, ops
    ops.set_compute_dtype(torch.bfloat16)
    pipe = _pipe()
    prompt = torch.randn(1, 77, 64).to(torch.bfloat16)
    tensors = _fake_s_tensors(pipe, prompt)
    path = str(tmp_path / "omgsr_constants.safetensors")
    K.write(path, tensors, K._meta("S", pipe, {}, {"unet": pipe.unet}))
    d = K.describe(path)
    assert d["metadata"]["format"] == "omgsr-constants" and d["metadata"]["family"] == "S" and d["metadata"]["tier"] == "bf16"
    assert d["metadata"]["mid_timestep"] == "273" and "checksum.unet" in d["metadata"] and set(d["tensors"]) == set(tensors)
    loaded_prompt = K.load_s(pipe, path)
    assert torch.equal(loaded_prompt, prompt)
    fb = pipe.unet._folded_biases(273)                 # served from the primed cache: the fake values, not a re-fold
    for i, r in enumerate(pipe.unet._resnets()):
        assert torch.equal(fb[id(r)], torch.full((r.conv1.out_channels,), float(i)))
    for n, at in K._s_cross_attention(pipe.unet):
        kk, vt, Lk = at._ctx_cache.get((loaded_prompt,), at._ctx_key(), lambda: pytest.fail("cache was not primed"))
        assert Lk == 77 and torch.equal(kk, tensors[f"unet.ctx.{n}.k"]) and torch.equal(vt, tensors[f"unet.ctx.{n}.vt"])
        # another prompt TENSOR (even with equal contents) is a different input: the builder runs
        with pytest.raises(RuntimeError, match="rebuilt"):
            at._ctx_cache.get((prompt.clone(),), at._ctx_key(), lambda: (_ for _ in ()).throw(RuntimeError("rebuilt")))


def test_a_file_from_another_deployment_is_refused(tmp_path):
    from omgsr_amd import constants as K, ops
    ops.set_compute_dtype(torch.bfloat16)
    pipe = _pipe()
    prompt = torch.randn(1, 77, 64).to(torch.bfloat16)
    path = str(tmp_path / "c.safetensors")
    K.write(path, _fake_s_tensors(pipe, prompt), K._meta("S", pipe, {}, {"unet": pipe.unet}))
    other_t = SimpleNamespace(unet=pipe.unet, mid_timestep=100)
    with pytest.raises(K.ConstantsMismatch, match="mid_timestep"):
        K.load_s(other_t, path)
    with torch.no_grad():
        pipe.unet.conv_in.weight[0, 0, 0, 0] += 1.0              # e.g. a LoRA merged after the export
    with pytest.raises(K.ConstantsMismatch, match="checksum.unet"):
        K.load_s(pipe, path)
    try:
        ops.set_compute_dtype(torch.float16)
        with pytest.raises(K.ConstantsMismatch, match="tier"):
            K.load_s(_pipe(), path)
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    bad = str(tmp_path / "bad.safetensors")
    K.write(bad, {"x": torch.zeros(1)}, {"format": "something-else"})
    with pytest.raises(K.ConstantsMismatch, match="not an omgsr-constants"):
        K.read(bad, "cpu")

"""The drop-in boundary, op by op: the reference's tiled VAE does not call `vae.encode`; it
duck-types the Encoder / Decoder attribute tree and calls every LEAF module itself, in NCHW, one tile at a time
(infer/vaehook.py:230-276 resblock2task / attn2task, :332-359 build_task_queue, :137-171 attn_forward_new, :384-413
custom_group_norm). This test restates that executor on the device (torch glue exactly where the reference has torch glue:
F.batch_norm GroupNorm with merged statistics, in-place SiLU, residual adds, crop / paste) and drives the PRODUCT modules
through it — `conv(x)`, `norm.weight / .bias`, `attn.to_q(...)`, `attn.prepare_attention_mask / head_to_batch_dim /
get_attention_scores / batch_to_head_dim`, `attn.to_out[0] / [1]` — then compares with the product's own fused tiled path
(omgsr_amd/pipelines/vaehook.py) and with the fp32 CPU oracle of the same algorithm (oracle/vaehook_ref.py, pinned to the
reference's outputs by tests/test_vaehook_golden.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
HOOK_VAE = dict(block_out_channels=[32, 32, 64, 64], layers_per_block=2, norm_num_groups=32)


def attn_forward_ref_style(attn, h_):
    """infer/vaehook.py:137-171, statement by statement, on the product attention module."""
    batch_size, channel, height, width = h_.shape
    hidden_states = h_.view(batch_size, channel, height * width).transpose(1, 2)
    attention_mask = attn.prepare_attention_mask(None, hidden_states.shape[1], batch_size)
    query = attn.to_q(hidden_states)
    key, value = attn.to_k(hidden_states), attn.to_v(hidden_states)
    query, key, value = attn.head_to_batch_dim(query), attn.head_to_batch_dim(key), attn.head_to_batch_dim(value)
    probs = attn.get_attention_scores(query, key, attention_mask)
    hidden_states = attn.batch_to_head_dim(torch.bmm(probs, value))
    hidden_states = attn.to_out[1](attn.to_out[0](hidden_states))
    return hidden_states.transpose(-1, -2).reshape(batch_size, channel, height, width)


def custom_group_norm(x, groups, mean, var, weight, bias, eps=1e-6):     # infer/vaehook.py:384-413
    b, c = x.shape[:2]
    xr = x.contiguous().view(1, b * groups, -1)
    out = F.batch_norm(xr, mean.to(x.dtype), var.to(x.dtype), weight=None, bias=None, training=False, momentum=0, eps=eps).view(x.shape)
    return out * weight.view(1, -1, 1, 1).to(x.dtype) + bias.view(1, -1, 1, 1).to(x.dtype)


def run_task_queue(net, x, tile_size, is_decoder):
    """Exact-mode VAEHook.vae_tile_forward (infer/vaehook.py:681-829) as a layer-synchronous loop over the task queue."""
    from oracle import vaehook_ref as V           # geometry + op order (pinned to the reference by the golden tests)
    pad = 11 if is_decoder else 32
    N, _, H, W = x.shape
    ins, outs = V.split_tiles(H, W, tile_size, pad, is_decoder)
    tiles = [x[:, :, b[2]:b[3], b[0]:b[1]].clone() for b in ins]
    ops_ = V.op_list(net, is_decoder)
    res = [[] for _ in tiles]
    for op in ops_:
        if op[0] == "gn":
            norm = op[1]
            stats = [V.group_var_mean(t.float(), norm.num_groups) for t in tiles]
            px = torch.tensor([t.shape[2] * t.shape[3] for t in tiles], dtype=torch.float32, device=x.device)
            p = (px / px.max()); p = (p / p.sum())[:, None]
            var, mean = (torch.vstack([s[0] for s in stats]) * p).sum(0), (torch.vstack([s[1] for s in stats]) * p).sum(0)
            tiles = [custom_group_norm(t, norm.num_groups, mean, var, norm.weight, norm.bias) for t in tiles]
            if op[2]:
                tiles = [F.silu(t, inplace=True) for t in tiles]
        elif op[0] == "f":
            tiles = [op[1](t) for t in tiles]
        elif op[0] == "res_push":
            for i, t in enumerate(tiles):
                res[i].append(op[1](t))
        else:
            tiles = [t + res[i].pop() for i, t in enumerate(tiles)]
    Ho, Wo = (H * 8, W * 8) if is_decoder else (H // 8, W // 8)
    result = torch.zeros((N, tiles[0].shape[1], Ho, Wo), dtype=torch.float32, device=x.device)
    for t, ib, ob in zip(tiles, ins, outs):
        pb = [v * 8 if is_decoder else v // 8 for v in ib]
        mg = [ob[i] - pb[i] for i in range(4)]
        result[:, :, ob[2]:ob[3], ob[0]:ob[1]] = t[:, :, mg[2]:t.shape[2] + mg[3], mg[0]:t.shape[3] + mg[1]]
    return result


@pytest.mark.parametrize("wd,tol", [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)], ids=["accurate", "bf16"])
def test_leaf_modules_drive_the_reference_task_queue(wd, tol, monkeypatch):
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL
    from omgsr_amd.pipelines.vaehook import VAEHook
    from omgsr_amd.testing import rel_l2, seeded_init_
    from oracle import diffusers_ref as R
    from oracle import vaehook_ref as V
    try:
        ops.set_compute_dtype(wd)
        o = seeded_init_(R.AutoencoderKL(**HOOK_VAE), 9).eval()
        p = AutoencoderKL(**HOOK_VAE)
        p.load_state_dict(o.state_dict())
        p = p.to(DEV, wd).eval()
        if wd == torch.float32:
            from omgsr_amd.precision import apply_default_policy
            apply_default_policy(vae=p)          # what OMGSR_{S,F}_Infer(weight_dtype=float32) installs
        g = torch.Generator().manual_seed(41)
        img = torch.randn(2, 3, 160, 224, generator=g).clamp(-2, 2).to(torch.bfloat16).float()
        z = torch.randn(2, 4, 28, 36, generator=g).to(torch.bfloat16).float()
        monkeypatch.setattr(V, "_attention_in_tile", attn_forward_ref_style)     # the op list calls the reference-style attention
        with torch.no_grad():
            ref_e = V.tiled_forward(o.encoder, img, 64, is_decoder=False)
            ref_d = V.tiled_forward(o.decoder, z, 12, is_decoder=True)
            leaf_e = run_task_queue(p.encoder, img.to(DEV, wd), 64, False)
            leaf_d = run_task_queue(p.decoder, z.to(DEV, wd), 12, True)
            p.encoder._tile_hook = VAEHook(p.encoder, 64, is_decoder=False, fast_decoder=False, fast_encoder=False, color_fix=False)
            p.decoder._tile_hook = VAEHook(p.decoder, 12, is_decoder=True, fast_decoder=False, fast_encoder=False, color_fix=False)
            fused_e, fused_d = p.encoder(img.to(DEV)), p.decoder(z.to(DEV))
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    for name, leaf, fused, ref in (("encoder", leaf_e, fused_e, ref_e), ("decoder", leaf_d, fused_d, ref_d)):
        e_ref, e_fused = rel_l2(leaf, ref), rel_l2(leaf, fused)
        print(f"{name} [{wd}]: leaf-by-leaf vs CPU oracle {e_ref:.2e}, vs the fused tiled path {e_fused:.2e}")
        assert leaf.shape == ref.shape and torch.isfinite(leaf).all()
        assert e_ref < tol and e_fused < tol


def test_attention_helper_surface_matches_torch():
    """prepare_attention_mask / head_to_batch_dim / get_attention_scores / batch_to_head_dim on [B, L, C] tensors."""
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api.autoencoder_kl import VaeAttention
    ops.set_compute_dtype(torch.bfloat16)
    at = VaeAttention(64, 32).to(DEV, torch.bfloat16)
    q = torch.randn(2, 150, 64, generator=torch.Generator().manual_seed(1)).to(DEV, torch.bfloat16)
    k = torch.randn(2, 150, 64, generator=torch.Generator().manual_seed(2)).to(DEV, torch.bfloat16)
    assert at.prepare_attention_mask(None, 150, 2) is None
    qb = at.head_to_batch_dim(q)
    assert qb.shape == (2, 150, 64) and torch.equal(at.batch_to_head_dim(qb), q)
    p = at.get_attention_scores(qb, at.head_to_batch_dim(k))
    ref = torch.softmax(torch.bmm(q.float(), k.float().transpose(1, 2)) * 64 ** -0.5, -1)
    assert p.dtype == torch.bfloat16 and p.shape == (2, 150, 150)
    assert (p.float() - ref).abs().max().item() < 4e-3 and torch.allclose(p.float().sum(-1), torch.ones(2, 150, device=DEV), atol=2e-2)

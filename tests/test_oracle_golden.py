"""Pins the oracle (oracle/pipeline_ref.py) and the product's host-side tiling logic against golden
vectors captured from the REFERENCE's own code (tests/golden/make_golden.py imported
infer/omgsr_{s,f}_infer_model.py and infer/infer_omgsr_f.py in the build container). CPU only."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import pipeline_ref as P

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "omgsr_pipeline.npz"))


def T(name):
    return torch.from_numpy(G[name])


# the same deterministic stand-ins the capture script used (test helpers, not reference code)
class FakeUNet:
    def __init__(self):
        self.calls = []

    def __call__(self, x, t, encoder_hidden_states=None):
        i = len(self.calls)
        self.calls.append(tuple(x.shape))
        y = 0.5 * x + 0.25 * torch.roll(x, 1, dims=-1) + 0.01 * (i + 1) + 0.001 * encoder_hidden_states.mean()
        return types.SimpleNamespace(sample=y)


class FakeVAE:
    def __init__(self, scaling_factor, shift_factor=None):
        self.config = types.SimpleNamespace(scaling_factor=scaling_factor, shift_factor=shift_factor, block_out_channels=[1, 2, 3, 4])
        self.seen = []

    def decode(self, z, return_dict=True):
        self.seen.append(z.clone())
        img = torch.nn.functional.interpolate(z[:, :3] * 1.5, scale_factor=2.0, mode="nearest")
        return types.SimpleNamespace(sample=img) if return_dict else (img,)


class FakeFlux:
    def __init__(self):
        self.calls = []

    def __call__(self, hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids, return_dict=False):
        i = len(self.calls)
        self.calls.append(tuple(hidden_states.shape))
        assert abs(float(timestep[0]) - float(G["t_curr"])) < 1e-7 and float(guidance[0]) == 1.0
        y = 0.5 * hidden_states - 0.125 * torch.roll(hidden_states, 3, dims=1) + 0.01 * (i + 1) + 0.001 * pooled_projections.mean()
        return (y,)


@pytest.mark.parametrize("tw,th", [(64, 64), (128, 128), (48, 32)])
def test_gaussian_weights(tw, th):
    ref = G[f"gauss_{tw}x{th}"]
    got = P.gaussian_weights(tw, th, 1, 4)
    assert got.dtype == torch.float64 and got.shape == (1, 4, th, tw)
    np.testing.assert_allclose(got[0, 0].numpy(), ref, rtol=1e-14, atol=0)
    # product host code (pure python/numpy, no GPU needed)
    from omgsr_amd.pipelines.latent_tiling import gaussian_weights
    np.testing.assert_allclose(gaussian_weights(tw, th), ref, rtol=1e-14, atol=0)
    if tw == th == 64:   # SURVEY C-2: symmetric left-right, NOT top-bottom
        assert np.abs(ref - ref[:, ::-1]).max() < 1e-12 and np.abs(ref - ref[::-1]).max() > 1.0


@pytest.mark.parametrize("name", ["s128", "s96", "s160x128", "s72x200", "s40x40_t16"])
def test_s_forward_tile(name):
    ts, ov = (int(v) for v in G[f"{name}_args"])
    lat = T(f"{name}_latent")
    unet, vae = FakeUNet(), FakeVAE(0.18215)
    m = P.OmgsrSRef(vae, unet, T("alpha_273"), 273)
    img = m.forward_tile(lat, T("s_ehs"), ts, ov)
    assert [list(c) for c in unet.calls] == G[f"{name}_calls"].tolist()          # call count, order and tile shapes
    torch.testing.assert_close(vae.seen[0], T(f"{name}_decoded_latent"), rtol=0, atol=0)
    if f"{name}_img" in G:
        torch.testing.assert_close(img, T(f"{name}_img"), rtol=0, atol=0)
    # product tile grid == the offsets implied by the reference run
    from omgsr_amd.pipelines.latent_tiling import tile_grid
    h, w = lat.shape[-2:]
    ets, offs = tile_grid(h, w, ts, ov)
    rts, _, _, roffs = P.tile_offsets(h, w, ts, ov)
    assert ets == rts and offs == roffs and len(offs) == len(unet.calls)


def test_s_forward_no_tile():
    unet, vae = FakeUNet(), FakeVAE(0.18215)
    m = P.OmgsrSRef(vae, unet, T("alpha_273"), 273)
    img = m.forward_no_tile(T("s_notile_latent"), T("s_ehs"))
    torch.testing.assert_close(vae.seen[0], T("s_notile_decoded_latent"), rtol=0, atol=0)
    torch.testing.assert_close(img, T("s_notile_img"), rtol=0, atol=0)


def test_alpha_and_schedule():
    from oracle.diffusers_ref import DDPMScheduler
    a = DDPMScheduler().alphas_cumprod[273]
    assert a.item() == float(G["alpha_273"]) == pytest.approx(0.6357423067092896, abs=1e-9)
    ts = P.flux_timesteps()
    np.testing.assert_array_equal(np.array(ts), G["flux_timesteps"])
    assert ts[-(244 + 1)] == float(G["t_curr"]) == pytest.approx(0.5051124691963196, abs=1e-12) and ts[-1] == 0.0
    # product
    from omgsr_amd.diffusers_api import DDPMScheduler as PS
    assert PS().alphas_cumprod[273].item() == a.item()


def test_pack_unpack_and_ids():
    x = T("pack_in")
    p = P.pack_latents(x, 2, 16, 12, 20)
    torch.testing.assert_close(p, T("pack_out"), rtol=0, atol=0)
    torch.testing.assert_close(P.unpack_latents(p, 96, 160, 8), T("unpack_out"), rtol=0, atol=0)
    torch.testing.assert_close(P.unpack_latents(p, 96, 160, 8), x, rtol=0, atol=0)
    torch.testing.assert_close(P.prepare_latent_image_ids(64, 64), T("ids_64x64"), rtol=0, atol=0)
    torch.testing.assert_close(P.prepare_latent_image_ids(6, 10), T("ids_6x10"), rtol=0, atol=0)
    assert T("ids_64x64")[65].tolist() == [0.0, 1.0, 1.0]


def _fref():
    vae, flux = FakeVAE(0.3611, 0.1159), FakeFlux()
    m = P.OmgsrFRef(vae, flux, 244, 1.0)
    assert m.t_curr == float(G["t_curr"]) and m.t_prev == 0.0 and m.vae_scale_factor == 8
    return m, vae, flux


def test_f_forward_no_tile():
    m, vae, _ = _fref()
    pe, tids = torch.zeros(1, 5, 32), torch.zeros(5, 3)
    img = m.forward_no_tile(T("f_notile_latent"), pe, T("f_pooled"), tids, P.prepare_latent_image_ids(16, 16))
    torch.testing.assert_close(vae.seen[0], T("f_notile_decoded_latent"), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(img, T("f_notile_img"), rtol=1e-6, atol=1e-6)


def test_f_forward_tile():
    m, vae, flux = _fref()
    pe, tids = torch.zeros(1, 5, 32), torch.zeros(5, 3)
    img = m.forward_tile(T("f_tile_latent"), pe, T("f_pooled"), tids, P.prepare_latent_image_ids(16, 16), 32, 16)
    assert [list(c) for c in flux.calls] == G["f_tile_calls"].tolist()
    torch.testing.assert_close(vae.seen[0], T("f_tile_decoded_latent"), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(img, T("f_tile_img"), rtol=1e-6, atol=1e-6)


def test_streamed_flux_equals_resident():
    """oracle/flux_streamed_ref.py (one block's weights resident at a time: the full-depth FLUX oracle) == the resident oracle
    model, bit for bit, on a small configuration."""
    from omgsr_amd.testing import seeded_init_
    from oracle import diffusers_ref as R
    from oracle.flux_streamed_ref import StreamedFlux, module_fetcher
    cfg = dict(num_layers=2, num_single_layers=3, attention_head_dim=16, num_attention_heads=2, joint_attention_dim=24,
               pooled_projection_dim=12, in_channels=8, axes_dims_rope=[4, 6, 6])
    m = seeded_init_(R.FluxTransformer2DModel(**cfg), 5).eval()
    g = torch.Generator().manual_seed(0)
    x, pe, pooled = torch.randn(2, 16, 8, generator=g), torch.randn(1, 5, 24, generator=g), torch.randn(1, 12, generator=g)
    tids, iids = torch.zeros(5, 3), P.prepare_latent_image_ids(4, 4)
    kw = dict(hidden_states=x, timestep=torch.tensor([0.505]), guidance=torch.ones(2), pooled_projections=pooled,
              encoder_hidden_states=pe, txt_ids=tids, img_ids=iids, return_dict=False)
    with torch.no_grad():
        ref = m(**kw)[0]
    st = StreamedFlux(module_fetcher(m), **cfg)
    st.trace = []
    got = st(**kw)[0]
    assert torch.equal(got, ref) and len(st.trace) == 1 and torch.equal(st.trace[0][1], ref)

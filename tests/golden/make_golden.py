"""Captures golden vectors by IMPORTING the reference's own pipeline code (build container only).

Run:  python tests/golden/make_golden.py      (needs /root/reference; never runs on the GPU box)

diffusers / peft / torchvision are not installed, so they are stubbed in sys.modules (the recipe of
SURVEY.md Appendix E); only the reference's own Python (tile grids, Gaussian stitch, latent algebra,
Flux pack/unpack, schedule, latent ids) executes. Outputs are numeric fixtures only:
tests/golden/omgsr_pipeline.npz (+ vaehook.npz from make_golden_vaehook.py). No reference source text
is stored.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present: golden capture only runs in the build container")
    sys.path.insert(0, REF)

    class _Any:
        pass
    _stub("diffusers", AutoencoderKL=_Any, UNet2DConditionModel=_Any, DDPMScheduler=_Any, FluxTransformer2DModel=_Any, FluxPipeline=_Any)
    _stub("diffusers.training_utils", free_memory=lambda: None)
    _stub("peft", PeftModel=_Any)
    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms", ToTensor=_Any, ToPILImage=_Any)
    tvf = _stub("torchvision.transforms.functional")
    tv.transforms, tvt.functional = tvt, tvf
    S = importlib.import_module("infer.omgsr_s_infer_model")
    Fm = importlib.import_module("infer.omgsr_f_infer_model")
    Fd = importlib.import_module("infer.infer_omgsr_f")
    return S, Fm, Fd


class FakeUNet:
    """Deterministic stand-in denoiser: depends on the tile CONTENT and the call index so that tile
    order, offsets and weighting are all pinned."""

    def __init__(self, in_channels=4):
        self.config = types.SimpleNamespace(in_channels=in_channels)
        self.dtype = torch.float32
        self.calls = []

    def __call__(self, x, t, encoder_hidden_states=None):
        i = len(self.calls)
        self.calls.append((tuple(x.shape), int(t)))
        y = 0.5 * x + 0.25 * torch.roll(x, 1, dims=-1) + 0.01 * (i + 1) + 0.001 * encoder_hidden_states.mean()
        return types.SimpleNamespace(sample=y)


class FakeVAE:
    def __init__(self, scaling_factor, shift_factor=None):
        self.config = types.SimpleNamespace(scaling_factor=scaling_factor, shift_factor=shift_factor, block_out_channels=[1, 2, 3, 4])
        self.dtype = torch.float32
        self.seen = []

    def decode(self, z, return_dict=True):
        self.seen.append(z.clone())
        img = torch.nn.functional.interpolate(z[:, :3] * 1.5, scale_factor=2.0, mode="nearest")
        return types.SimpleNamespace(sample=img) if return_dict else (img,)


class FakeFlux:
    dtype = torch.float32

    def __init__(self):
        self.calls = []

    def __call__(self, hidden_states, timestep, guidance, pooled_projections, encoder_hidden_states, txt_ids, img_ids, return_dict=False):
        i = len(self.calls)
        self.calls.append((tuple(hidden_states.shape), float(timestep[0]), float(guidance[0])))
        y = 0.5 * hidden_states - 0.125 * torch.roll(hidden_states, 3, dims=1) + 0.01 * (i + 1) + 0.001 * pooled_projections.mean()
        return (y,)


def main():
    S, Fm, Fd = import_reference()
    g = torch.Generator().manual_seed(20260101)
    out = {}

    # G1: Gaussian weights
    m = S.OMGSR_S_Infer.__new__(S.OMGSR_S_Infer)
    torch.nn.Module.__init__(m)
    m.device = "cpu"
    m.unet = FakeUNet()
    for (tw, th) in [(64, 64), (128, 128), (48, 32)]:
        out[f"gauss_{tw}x{th}"] = m._gaussian_weights(tw, th, 1)[0, 0].numpy()

    # G2/G3: OMGSR-S tile + no-tile forward with a deterministic fake UNet / VAE
    alpha = torch.cumprod(1.0 - torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2, 0)[273]
    out["alpha_273"] = alpha.numpy()
    ehs = torch.randn(1, 7, 16, generator=g)
    out["s_ehs"] = ehs.numpy()
    for name, (B, h, w, ts, ov) in {"s128": (1, 128, 128, 64, 32), "s96": (1, 96, 96, 64, 32), "s160x128": (1, 160, 128, 64, 32),
                                    "s72x200": (1, 72, 200, 64, 32), "s40x40_t16": (1, 40, 40, 16, 8)}.items():
        m.unet, m.vae, m.mid_timestep, m.alpha_t = FakeUNet(), FakeVAE(0.18215), 273, alpha
        lat = torch.randn(B, 4, h, w, generator=g)
        img = m._forward_tile(lat, ehs, ts, ov)
        out[f"{name}_latent"] = lat.numpy()
        if h * w <= 96 * 96:     # the image is a fixed function of the decoded latent in FakeVAE: keep fixtures small
            out[f"{name}_img"] = img.numpy()
        out[f"{name}_decoded_latent"] = m.vae.seen[0].numpy()
        out[f"{name}_calls"] = np.array([c[0] for c in m.unet.calls])
        out[f"{name}_args"] = np.array([ts, ov])
    m.unet, m.vae = FakeUNet(), FakeVAE(0.18215)
    lat = torch.randn(2, 4, 64, 64, generator=g)
    out["s_notile_latent"] = lat.numpy()
    out["s_notile_img"] = m._forward_no_tile(lat, ehs).numpy()
    out["s_notile_decoded_latent"] = m.vae.seen[0].numpy()

    # G4: pack / unpack
    x = torch.randn(2, 16, 12, 20, generator=g)
    p = Fm._pack_latents(x, 2, 16, 12, 20)
    out["pack_in"], out["pack_out"] = x.numpy(), p.numpy()
    out["unpack_out"] = Fm._unpack_latents(p, 12 * 8, 20 * 8, 8).numpy()

    # G5: schedule
    ts_all = Fm.get_flux_setting_timesteps()
    out["flux_timesteps"] = np.array(ts_all, dtype=np.float64)

    # G6: latent image ids
    out["ids_64x64"] = Fd._prepare_latent_image_ids(64, 64, "cpu", torch.float32).numpy()
    out["ids_6x10"] = Fd._prepare_latent_image_ids(6, 10, "cpu", torch.float32).numpy()

    # OMGSR-F forward algebra (tile + no tile) with fakes
    f = Fm.OMGSR_F_Infer.__new__(Fm.OMGSR_F_Infer)
    torch.nn.Module.__init__(f)
    f.device, f.weight_dtype, f.guidance_scale, f.mid_timestep = "cpu", torch.float32, 1.0, 244
    f.vae_scale_factor = 8
    f.t_curr, f.t_prev = ts_all[-(244 + 1)], ts_all[-1]
    out["t_curr"], out["t_prev"] = np.float64(f.t_curr), np.float64(f.t_prev)
    pe, pooled = torch.randn(1, 5, 32, generator=g), torch.randn(1, 24, generator=g)
    out["f_pooled"] = pooled.numpy()
    tids = torch.zeros(5, 3)
    f.vae, f.flux_transformer = FakeVAE(0.3611, 0.1159), FakeFlux()
    lat = torch.randn(2, 16, 32, 32, generator=g)
    iids = Fd._prepare_latent_image_ids(16, 16, "cpu", torch.float32)
    out["f_notile_latent"] = lat.numpy()
    out["f_notile_img"] = f._forward_no_tile(lat, pe, pooled, tids, iids).numpy()
    out["f_notile_decoded_latent"] = f.vae.seen[0].numpy()
    f.vae, f.flux_transformer = FakeVAE(0.3611, 0.1159), FakeFlux()
    lat = torch.randn(1, 16, 48, 40, generator=g)
    out["f_tile_latent"] = lat.numpy()
    out["f_tile_img"] = f._forward_tile(lat, pe, pooled, tids, iids, 32, 16).numpy()
    out["f_tile_decoded_latent"] = f.vae.seen[0].numpy()
    out["f_tile_calls"] = np.array([c[0] for c in f.flux_transformer.calls])

    path = os.path.join(HERE, "omgsr_pipeline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()

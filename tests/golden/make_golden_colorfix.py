"""Captures golden vectors for the colour-fix post-process by IMPORTING the reference's infer/wavelet_color_fix.py
(build container only; torchvision / PIL are absent and stubbed, so only its tensor-level functions run).

Run:  python tests/golden/make_golden_colorfix.py   ->  tests/golden/colorfix.npz   (numeric fixtures only)
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


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present: golden capture only runs in the build container")
    sys.path.insert(0, REF)

    class _Any:
        pass
    _stub("PIL", Image=_Any)
    _stub("PIL.Image", Image=_Any)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", ToTensor=_Any, ToPILImage=_Any)
    W = importlib.import_module("infer.wavelet_color_fix")

    g = torch.Generator().manual_seed(4321)
    H, Wd = 44, 60
    # uint8 "images": a smooth field + noise (target = SR output, source = upscaled LQ with a colour cast)
    base = torch.nn.functional.interpolate(torch.rand(1, 3, 6, 8, generator=g), size=(H, Wd), mode="bicubic", align_corners=False)
    tgt = (base + 0.08 * torch.randn(1, 3, H, Wd, generator=g)).clamp(0, 1).mul(255).to(torch.uint8)
    src = (base * torch.tensor([0.8, 1.0, 1.15]).view(1, 3, 1, 1) + 0.05 + 0.02 * torch.randn(1, 3, H, Wd, generator=g)).clamp(0, 1).mul(255).to(torch.uint8)
    t32, s32 = tgt.float() / 255, src.float() / 255
    out = {"target_u8": tgt.numpy(), "source_u8": src.numpy()}
    m, s = W.calc_mean_std(t32)
    out["target_mean"], out["target_std"] = m.numpy(), s.numpy()
    out["adain"] = W.adaptive_instance_normalization(t32, s32).numpy()
    for r in (1, 4, 16):
        out[f"blur_r{r}"] = W.wavelet_blur(t32, r).numpy()
    hi, lo = W.wavelet_decomposition(t32)
    out["decomp_high"], out["decomp_low"] = hi.numpy(), lo.numpy()
    out["wavelet"] = W.wavelet_reconstruction(t32, s32).numpy()
    np.savez_compressed(os.path.join(HERE, "colorfix.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

"""SURVEY §8(f) f2 on the GPU: omgsr_resample_u8 / omgsr_amd.preprocess against Pillow's own outputs (tests/golden/pil_resize.npz)
and, at the sizes of the real workload (256 -> 1024), against the oracle restatement pinned to them — bit for bit on uint8."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pil_resize.npz"))
CASES = sorted(k[:-3] for k in G.files if k.endswith(".in") and not k.startswith("chain"))


@pytest.mark.parametrize("name", CASES)
def test_resize_u8_matches_pillow_golden(name):
    from omgsr_amd import preprocess as PP
    ow, oh, f = (int(v) for v in G[name + ".meta"])
    x = torch.from_numpy(G[name + ".in"])[None].to(DEV)
    got = PP.resize_u8(x, (ow, oh), PP.BICUBIC if f == 0 else PP.LANCZOS)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (1, oh, ow, 3)
    assert np.array_equal(got[0].cpu().numpy(), G[name + ".out"])


def test_driver_chain_matches_pillow_golden():
    from omgsr_amd import preprocess as PP
    x = torch.from_numpy(G["chain.in"])[None].to(DEV)
    got = PP.preprocess_u8(x, process_size=160, upscale=4)
    assert np.array_equal(got[0].cpu().numpy(), G["chain.out"])


def test_workload_size_batch_matches_oracle():
    """256x256 -> 1024x1024 (BASELINE configs[2]) and a ragged 250x333 input (x4 = 1000 x 1332 -> snap to 1000 x 1328), batch 3."""
    from omgsr_amd import preprocess as PP
    from omgsr_amd.colorfix import image_to_model_input
    from oracle import pil_resize_ref as P
    rng = np.random.default_rng(7)
    for h, w in ((256, 256), (250, 333)):
        imgs = rng.integers(0, 256, size=(3, h, w, 3), dtype=np.uint8)
        got = PP.preprocess_u8(torch.from_numpy(imgs).to(DEV), 512, 4)
        ref = np.stack([P.driver_preprocess(im, 512, 4) for im in imgs])
        assert got.shape[1] % 8 == 0 and got.shape[2] % 8 == 0 and np.array_equal(got.cpu().numpy(), ref)
    x = image_to_model_input(got)
    assert x.shape[-1] == 8 and float(x.float().abs().max()) <= 1.0


def test_identity_and_errors():
    from omgsr_amd import preprocess as PP
    x = torch.randint(0, 256, (2, 16, 24, 3), dtype=torch.uint8, device=DEV)
    y = PP.resize_u8(x, (24, 16))
    assert torch.equal(x, y) and y.data_ptr() != x.data_ptr()
    with pytest.raises(ValueError):
        PP.resize_u8(x, (0, 16))
    with pytest.raises(ValueError):
        PP.resize_u8(x, (25, 16), "nearest")

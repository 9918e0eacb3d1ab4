"""Pins oracle/vaehook_ref.py (and the product's tile-split host logic) against golden vectors produced
by the REFERENCE's own infer/vaehook.py (tests/golden/make_golden_vaehook.py). CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusers_ref as R
from oracle import vaehook_ref as V

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vaehook.npz"))
SPLITS = sorted(k[len("split_"):-len("_args")] for k in G.files if k.endswith("_args"))


def T(name):
    return torch.from_numpy(G[name])


@pytest.mark.parametrize("name", SPLITS)
def test_split_tiles(name):
    is_dec, tile, h, w = (int(v) for v in G[f"split_{name}_args"])
    pad = 11 if is_dec else 32
    ins, outs = V.split_tiles(h, w, tile, pad, bool(is_dec))
    assert ins == G[f"split_{name}_in"].tolist() and outs == G[f"split_{name}_out"].tolist()
    from omgsr_amd.pipelines.vaehook import split_tiles
    pins, pouts = split_tiles(h, w, tile, pad, bool(is_dec))
    assert pins == ins and pouts == outs


def test_known_geometry():
    """SURVEY Appendix E observations."""
    ins, outs = V.split_tiles(1024, 1024, 256, 32, False)
    assert len(ins) == 16 and outs[0] == [0, 36, 0, 36]
    assert sorted({b[1] - b[0] for b in ins}) == [256, 320]
    ins, outs = V.split_tiles(128, 128, 64, 11, True)
    assert len(ins) == 4 and outs[0] == [0, 600, 0, 600] and sorted({b[1] - b[0] for b in ins}) == [64, 86]


def test_group_stat_helpers():
    t1, t2 = T("gn_t1"), T("gn_t2")
    v1, m1 = V.group_var_mean(t1)
    torch.testing.assert_close(v1, T("gn_var1"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(m1, T("gn_mean1"), rtol=1e-5, atol=1e-6)
    v2, m2 = V.group_var_mean(t2)
    var, mean = V.merge_stats([v1, v2], [m1, m2], [t1.shape[2] * t1.shape[3], t2.shape[2] * t2.shape[3]])
    y = V.fixed_group_norm(t1, mean, var, T("gn_w"), T("gn_b"))
    torch.testing.assert_close(y, T("gn_merged_t1"), rtol=1e-4, atol=1e-5)


def _vae():
    from omgsr_amd.testing import seeded_init_
    return seeded_init_(R.AutoencoderKL(block_out_channels=[32, 32, 64, 64], layers_per_block=2, norm_num_groups=32), 9).eval()


@pytest.mark.parametrize("fast", [False, True])
def test_tiled_encoder_decoder_match_reference(fast):
    vae = _vae()
    tag = "fast" if fast else "exact"
    enc = V.tiled_forward(vae.encoder, T("enc_in"), 64, is_decoder=False, fast=fast)
    dec = V.tiled_forward(vae.decoder, T("dec_in"), 12, is_decoder=True, fast=fast)
    assert enc.dtype == torch.float32 and dec.dtype == torch.float32          # SURVEY C-10
    torch.testing.assert_close(enc, T(f"enc_out_{tag}"), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(dec[..., ::2, ::2], T(f"dec_out_{tag}_s2"), rtol=2e-4, atol=2e-4)


def test_untiled_when_small():
    vae = _vae()
    z = torch.randn(1, 4, 20, 30)
    with torch.no_grad():
        torch.testing.assert_close(V.tiled_forward(vae.decoder, z, 12, True), vae.decoder(z))     # max(H,W) <= 2*11+12

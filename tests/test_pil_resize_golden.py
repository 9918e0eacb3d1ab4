"""SURVEY §8(f) f2 — the oracle's restatement of Pillow's 8-bit resampler (oracle/pil_resize_ref.py) against outputs of Pillow
itself (tests/golden/pil_resize.npz, tests/golden/make_golden_pil_resize.py): bit for bit, every case; and the host-side
coefficient tables the HIP kernel consumes (omgsr_amd/preprocess.py) against the oracle's."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pil_resize.npz"))
CASES = sorted(k[:-3] for k in G.files if k.endswith(".in") and not k.startswith("chain"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_resize_matches_pillow(name):
    from oracle import pil_resize_ref as P
    ow, oh, f = (int(v) for v in G[name + ".meta"])
    got = P.resize(G[name + ".in"], (ow, oh), P.BICUBIC if f == 0 else P.LANCZOS)
    assert got.dtype == np.uint8 and got.shape == G[name + ".out"].shape
    assert np.array_equal(got, G[name + ".out"])


def test_oracle_driver_chain_matches_pillow():
    from oracle import pil_resize_ref as P
    got = P.driver_preprocess(G["chain.in"], process_size=160, upscale=4)
    assert got.shape[0] % 8 == 0 and got.shape[1] % 8 == 0 and np.array_equal(got, G["chain.out"])


@pytest.mark.parametrize("n_in,n_out,filt", [(31, 124, "bicubic"), (125, 120, "lanczos"), (40, 301, "bicubic"), (80, 30, "lanczos"), (7, 7 * 4, "bicubic")])
def test_product_coefficient_tables_equal_the_oracles(n_in, n_out, filt):
    """The product computes the fixed-point taps on the host (float64, like Pillow's C double code) and ships them to the kernel."""
    from omgsr_amd import preprocess as PP
    from oracle import pil_resize_ref as P
    b, kk, ks = P.precompute_coeffs(n_in, n_out, filt)
    pb, pkk, pks = PP.precompute_coeffs(n_in, n_out, filt)
    assert ks == pks and np.array_equal(b, pb) and np.array_equal(kk, pkk)

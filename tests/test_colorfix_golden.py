"""The colour-fix oracle (oracle/colorfix_ref.py) against vectors captured from the reference's own
infer/wavelet_color_fix.py (tests/golden/make_golden_colorfix.py). CPU only."""
import os

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "colorfix.npz"))


def t(name):
    return torch.from_numpy(G[name])


def test_calc_mean_std_and_adain_match_the_reference():
    from oracle import colorfix_ref as R
    tgt, src = R.to_tensor_f32(t("target_u8")), R.to_tensor_f32(t("source_u8"))
    m, s = R.calc_mean_std(tgt)
    assert torch.equal(m, t("target_mean")) and torch.equal(s, t("target_std"))
    assert torch.equal(R.adaptive_instance_normalization(tgt, src), t("adain"))


def test_wavelet_blur_decomposition_reconstruction_match_the_reference():
    from oracle import colorfix_ref as R
    tgt, src = R.to_tensor_f32(t("target_u8")), R.to_tensor_f32(t("source_u8"))
    for r in (1, 4, 16):
        assert torch.equal(R.wavelet_blur(tgt, r), t(f"blur_r{r}"))
    hi, lo = R.wavelet_decomposition(tgt)
    assert torch.equal(hi, t("decomp_high")) and torch.equal(lo, t("decomp_low"))
    assert torch.equal(R.wavelet_reconstruction(tgt, src), t("wavelet"))


def test_uint8_conversions():
    from oracle import colorfix_ref as R
    x = torch.tensor([-1.5, -1.0, 0.0, 0.999, 1.0, 1.2])
    assert R.model_output_to_u8(x).tolist() == [0, 0, 127, 254, 255, 255]        # truncation, not rounding
    u = torch.arange(256, dtype=torch.uint8)
    assert torch.equal(R.lq_to_u8(R.to_tensor_f32(u) * 2 - 1), u)
    # bf16 model output: the "+ 0.5" rounds to bf16 before the byte conversion (infer/infer_omgsr_s.py:96)
    xb = torch.tensor([0.3], dtype=torch.bfloat16)
    assert R.model_output_to_u8(xb).item() == int(float((xb * 0.5 + 0.5).float()) * 255)

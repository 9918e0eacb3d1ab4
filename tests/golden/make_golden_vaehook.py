"""Golden vectors of the reference's tiled VAE (infer/vaehook.py) — build container only.

Runs the REFERENCE's VAEHook / split_tiles / get_var_mean / custom_group_norm / GroupNormParam (imported
from /root/reference) on the oracle's reduced Encoder / Decoder (seeded weights) and stores inputs + outputs
in tests/golden/vaehook.npz. Only numbers are stored."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present")
    sys.path.insert(0, REF)
    import infer.devices as d
    import infer.vaehook as vh
    d.device = torch.device("cpu")          # else GroupNormParam.summary() asks for a GPU
    from oracle import diffusers_ref as R
    from omgsr_amd.testing import seeded_init_

    out = {}
    # G7: split_tiles
    for name, (is_dec, tile, h, w) in {"enc256_1024": (False, 256, 1024, 1024), "enc512_1024": (False, 512, 1024, 1024),
                                       "dec64_128": (True, 64, 128, 128), "dec32_64": (True, 32, 64, 64),
                                       "enc256_512": (False, 256, 512, 512), "enc96_200x312": (False, 96, 200, 312),
                                       "dec24_40x56": (True, 24, 40, 56)}.items():
        hook = vh.VAEHook(None, tile, is_decoder=is_dec, fast_decoder=False, fast_encoder=False, color_fix=False)
        ins, outs = hook.split_tiles(h, w)
        out[f"split_{name}_in"], out[f"split_{name}_out"] = np.array(ins), np.array(outs)
        out[f"split_{name}_args"] = np.array([int(is_dec), tile, h, w])

    # G9: statistics helpers
    g = torch.Generator().manual_seed(77)
    t1, t2 = torch.randn(2, 64, 9, 7, generator=g) * 2 + 0.5, torch.randn(2, 64, 5, 11, generator=g) - 1.0
    gp = vh.GroupNormParam()
    norm = torch.nn.GroupNorm(32, 64)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(64, generator=g)); norm.bias.copy_(torch.randn(64, generator=g))
    gp.add_tile(t1, norm); gp.add_tile(t2, norm)
    fn = gp.summary()
    v1, m1 = vh.get_var_mean(t1, 32)
    with torch.no_grad():
        out["gn_t1"], out["gn_t2"] = t1.numpy(), t2.numpy()
        out["gn_w"], out["gn_b"] = norm.weight.detach().numpy(), norm.bias.detach().numpy()
        out["gn_var1"], out["gn_mean1"] = v1.numpy(), m1.numpy()
        out["gn_merged_t1"] = fn(t1.clone()).numpy()

    # G8: VAEHook end to end on the oracle's reduced nets
    cfg = dict(block_out_channels=[32, 32, 64, 64], layers_per_block=2, norm_num_groups=32)   # the hook hard-codes 2 (+1) resnets per block
    vae = seeded_init_(R.AutoencoderKL(**cfg), 9).eval()
    enc, dec = vae.encoder, vae.decoder
    enc.original_forward, dec.original_forward = enc.forward, dec.forward
    img = torch.randn(1, 3, 160, 224, generator=g).clamp(-2, 2)
    z = torch.randn(2, 4, 28, 36, generator=g)
    out["enc_in"], out["dec_in"] = img.numpy(), z.numpy()
    with torch.no_grad():
        for fast in (False, True):
            he = vh.VAEHook(enc, 64, is_decoder=False, fast_decoder=fast, fast_encoder=fast, color_fix=False)
            hd = vh.VAEHook(dec, 12, is_decoder=True, fast_decoder=fast, fast_encoder=fast, color_fix=False)
            oe, od = he(img), hd(z)
            assert oe.dtype == torch.float32 and od.dtype == torch.float32
            tag = "fast" if fast else "exact"
            # decoder outputs are stored at every 2nd pixel (fixture size); the encoder output in full
            out[f"enc_out_{tag}"], out[f"dec_out_{tag}_s2"] = oe.numpy(), od[..., ::2, ::2].contiguous().numpy()
    path = os.path.join(HERE, "vaehook.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()

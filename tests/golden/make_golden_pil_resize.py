"""Golden vectors for the PIL-exact resampler: outputs of Pillow's own Image.resize — the library the
reference's driver calls (infer/infer_omgsr_s.py:71-84) — on seeded random RGB images. Run in the build container (Pillow is
installed here; the GPU box only reads the committed .npz):

    python tests/golden/make_golden_pil_resize.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
# (name, input H, W, output W, H, filter): the reference's three resizes at small sizes, ragged sizes, both passes / one pass
CASES = [
    ("x4_bicubic", 24, 31, 124, 96, Image.BICUBIC),
    ("x4_bicubic_sq", 32, 32, 128, 128, Image.BICUBIC),
    ("small_up_bicubic", 17, 40, 301, 128, Image.BICUBIC),       # the `ori < process_size // rscale` branch: non-integer scale
    ("snap8_lanczos", 100, 124, 120, 96, Image.LANCZOS),         # w - w % 8, h - h % 8
    ("snap8_lanczos_w_only", 96, 125, 120, 96, Image.LANCZOS),   # height already a multiple of 8: horizontal pass only
    ("snap8_lanczos_h_only", 99, 120, 120, 96, Image.LANCZOS),
    ("down_lanczos", 64, 80, 30, 24, Image.LANCZOS),             # scale > 1: the support widens
    ("down_bicubic", 50, 70, 33, 21, Image.BICUBIC),
]


def main():
    rng = np.random.default_rng(20260513)
    out = {}
    for name, h, w, ow, oh, filt in CASES:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        if "bicubic" in name:                       # smooth-ish content next to noise: overshoot clipping both ways
            img[: h // 2] = np.clip(np.cumsum(rng.integers(-9, 10, size=(h // 2, w, 3)), axis=1) + 128, 0, 255).astype(np.uint8)
        res = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), filt))
        out[f"{name}.in"], out[f"{name}.out"] = img, res
        out[f"{name}.meta"] = np.array([ow, oh, 0 if filt == Image.BICUBIC else 1], np.int32)
    # the driver's whole chain on one image (default resample of Image.resize for RGB is BICUBIC)
    # process_size 160 instead of the launcher's 512 keeps the fixture small; the arithmetic is the same
    img = rng.integers(0, 256, size=(29, 37, 3), dtype=np.uint8)
    pil = Image.fromarray(img, "RGB")
    scale = (160 // 4) / min(pil.size)
    pil = pil.resize((int(scale * pil.size[0]), int(scale * pil.size[1])))
    pil = pil.resize((pil.size[0] * 4, pil.size[1] * 4))
    pil = pil.resize((pil.width - pil.width % 8, pil.height - pil.height % 8), Image.LANCZOS)
    out["chain.in"], out["chain.out"] = img, np.asarray(pil)
    import PIL
    out["pillow_version"] = np.array([int(v) for v in PIL.__version__.split(".")[:2]], np.int32)
    np.savez_compressed(os.path.join(HERE, "pil_resize.npz"), **out)
    print("wrote", os.path.join(HERE, "pil_resize.npz"), {k: v.shape for k, v in out.items() if k.endswith(".out")})


if __name__ == "__main__":
    main()

"""A/B of the range-fallback tier (fp32 stream, bf16 operands, every operand and weight split) on OMGSR-S 128->512 at SD2.1 shapes against the fp32 CPU oracle:
the flash kernel with single q / k / P vs their two-term splits (ops.attn_split, OMGSR_ATTN_SPLIT), next to the accurate tier. GPU box:
    python tools/fallback_ab.py [weight seed] [rounded 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omgsr_amd import ops                                                      # noqa: E402
from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel        # noqa: E402
from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer                          # noqa: E402
from omgsr_amd.testing import psnr, rel_l2, seeded_init_, synthetic_lq         # noqa: E402
from oracle import diffusers_ref as R                                          # noqa: E402
from oracle.pipeline_ref import OmgsrSRef                                      # noqa: E402

wseed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounded = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
torch.set_num_threads(16)
vae = seeded_init_(R.AutoencoderKL(), 101 + wseed, rounded=rounded).eval()
unet = seeded_init_(R.UNet2DConditionModel(), 202 + wseed, rounded=rounded).eval()
g = torch.Generator().manual_seed(4321)
prompt = torch.randn(1, 77, 1024, generator=g)
x = synthetic_lq(1, 512, 512, seed=1234)
eps = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(99))
vae.posterior_noise = eps
with torch.no_grad():
    ref = OmgsrSRef(vae, unet, R.DDPMScheduler().alphas_cumprod[273], 273)(x, prompt, 64, 32)
for name, env, fallback in (("accurate", None, False), ("fallback, single q/k/P", "0", True), ("fallback, split q/k/P", "1", True)):
    if env is None:
        os.environ.pop("OMGSR_ATTN_SPLIT", None)
    else:
        os.environ["OMGSR_ATTN_SPLIT"] = env
    pv, pu = AutoencoderKL(), UNet2DConditionModel()
    pv.load_state_dict(vae.state_dict()); pu.load_state_dict(unet.state_dict())
    pipe = OMGSR_S_Infer(None, None, 273, "cuda", torch.float32, vae=pv, unet=pu)
    if fallback:
        pipe.range_fallback.enter()
    pipe.vae.posterior_noise = eps.to("cuda")
    with torch.no_grad():
        got, secs = pipe(x.to("cuda"), prompt.to("cuda"), 64, 32)
        got, secs = pipe(x.to("cuda"), prompt.to("cuda"), 64, 32)
    got = got.float().cpu()
    print(f"{name:28s} rel-L2 {rel_l2(got, ref):.3e}  PSNR {psnr(got, ref):.1f} dB  {secs * 1e3:.1f} ms  attn_split={ops.attn_split()}", flush=True)
    pipe.range_fallback.reset()
    del pipe, pv, pu
    torch.cuda.empty_cache()
    ops.set_compute_dtype(torch.bfloat16)

"""Find the FIRST op whose sample-0 result differs between B = 1 and B = 4 in the Flux DiT (same tokens in every batch entry).
GPU box only. python tools/flux_batch_trace.py [fp32|bf16]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omgsr_amd import ops  # noqa: E402
from omgsr_amd.diffusers_api import FluxTransformer2DModel  # noqa: E402
from omgsr_amd.pipelines.omgsr_f import get_flux_setting_timesteps, prepare_latent_image_ids  # noqa: E402
from omgsr_amd.testing import seeded_init_device_  # noqa: E402

tier = sys.argv[1] if len(sys.argv) > 1 else "fp32"
wd = torch.float32 if tier == "fp32" else torch.bfloat16
ops.set_compute_dtype(wd)
dev = "cuda"
with torch.device("meta"):
    f = FluxTransformer2DModel(num_layers=1, num_single_layers=1)
f = f.to_empty(device=dev)
seeded_init_device_(f, 404)
f = f.to(wd).eval()
if tier == "fp32":
    from omgsr_amd.precision import apply_default_policy
    apply_default_policy(flux=f)
g = torch.Generator().manual_seed(1)
tok = torch.randn(1, 4096, 64, generator=g).to(dev, wd)
pe, pooled = torch.randn(1, 512, 4096, generator=g).to(dev, wd), torch.randn(1, 768, generator=g).to(dev, wd)
tids, iids = torch.zeros(512, 3, device=dev, dtype=wd), prepare_latent_image_ids(64, 64, dev, wd)
t = torch.tensor([get_flux_setting_timesteps()[-(244 + 1)]], device=dev)

LOG = []
CUR_B = [1]


def wrap(name, fn, out_arg=None):
    def w(*a, **k):
        r = fn(*a, **k)
        out = r if out_arg is None else (k.get(out_arg) if out_arg in k else a[out_arg] if isinstance(out_arg, int) else None)
        if torch.is_tensor(out):
            B = CUR_B[0]
            o = out.float()
            o = o.reshape(B, -1)[0] if o.numel() % B == 0 else o.reshape(-1)
            LOG.append((name, tuple(out.shape), o.clone()))
        return r
    return w


ops.layer_norm = wrap("layer_norm", ops.layer_norm)
ops.linear = wrap("linear", ops.linear)
ops.linear_rows = wrap("linear_rows", ops.linear_rows)
ops.linear_into = wrap("linear_into", ops.linear_into, out_arg=2)
ops.linear_t_into = wrap("linear_t_into", ops.linear_t_into, out_arg=2)
ops.rmsnorm_rope_ = wrap("rmsnorm_rope_", ops.rmsnorm_rope_, out_arg=0)
_att = ops.attention


def att(*a, **k):
    r = _att(*a, **k)
    B = CUR_B[0]
    LOG.append(("attention", tuple(r.shape), r.float().reshape(B, -1)[0].clone()))
    return r


ops.attention = att


def fwd(B):
    CUR_B[0] = B
    LOG.clear()
    with torch.no_grad():
        out = f(hidden_states=tok.expand(B, -1, -1).contiguous(), timestep=t, guidance=torch.full((B,), 1.0, device=dev), pooled_projections=pooled,
                encoder_hidden_states=pe, txt_ids=tids, img_ids=iids, return_dict=False)[0].float()
    return out, list(LOG)


fwd(1); fwd(4)                       # warm caches
o1, l1 = fwd(1)
o4, l4 = fwd(4)
print(len(l1), len(l4))
for (n1, s1, a), (n4, s4, b) in zip(l1, l4):
    if a.numel() != b.numel():
        # joint buffers: B=1 [1, L, X] vs B=4 [4, L, X] -> same per-sample size; otherwise report and continue
        print(f"{n1:14s} {s1} vs {s4}: sizes differ ({a.numel()} vs {b.numel()})")
        continue
    d = (a - b).norm() / a.norm().clamp_min(1e-30)
    print(f"{n1:14s} {str(s1):28s} {str(s4):28s} rel diff {d.item():.3e}  max abs {float((a - b).abs().max()):.3e}")
print("final", ((o1 - o4[:1]).norm() / o1.norm()).item())

"""Why does OMGSR-F batch 8 differ from batch 1 by ~4e-4 (accurate tier) when both sit ~5.5e-4 from the oracle
(tests/test_flux_fullsize_gpu.py, round 4)? FLUX.1-dev width, 2 + 2 blocks, the DiT alone: B = 1 vs B = 4 (the same tokens in every
batch entry) under the dispatch switches that change between the two batch sizes. GPU box only.
    python tools/flux_batch_diff.py [fp32|bf16]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(tier):
    import torch
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import get_flux_setting_timesteps, prepare_latent_image_ids
    from omgsr_amd.testing import rel_l2, seeded_init_device_
    wd = torch.float32 if tier == "fp32" else torch.bfloat16
    ops.set_compute_dtype(wd)
    if os.environ.get("OMGSR_BI") == "1":
        ops.set_batch_invariant(True)
    dev = "cuda"
    with torch.device("meta"):
        f = FluxTransformer2DModel(num_layers=2, num_single_layers=2)
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

    def fwd(B):
        with torch.no_grad():
            return f(hidden_states=tok.expand(B, -1, -1).contiguous(), timestep=t, guidance=torch.full((B,), 1.0, device=dev), pooled_projections=pooled,
                     encoder_hidden_states=pe, txt_ids=tids, img_ids=iids, return_dict=False)[0].float()
    o1, o4 = fwd(1), fwd(4)
    print(f"  B=4[0] vs B=1: {rel_l2(o4[:1], o1):.3e}   B=4[0] vs B=4[3]: {rel_l2(o4[:1], o4[3:]):.3e}   bit-equal {torch.equal(o4[:1], o1)}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        run(sys.argv[1])
        sys.exit(0)
    tier = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    for name, env in [("default", {}), ("batch-invariant", {"OMGSR_BI": "1"}), ("no p8", {"OMGSR_P8": "0"}), ("dma only", {"OMGSR_IGEMM_MODE": "dma"}),
                      ("register-staged only", {"OMGSR_IGEMM_MODE": "reg"}), ("register-staged attention tiles", {"OMGSR_ATTN_VARIANT": "0"})]:
        print(f"{tier} {name}: {env}", flush=True)
        e = dict(os.environ, **env)
        subprocess.run([sys.executable, os.path.abspath(__file__), tier, "child"], env=e, check=False)

"""hipGraph replay of the pipelines' forward() (omgsr_amd/pipelines/graphed.py): a replayed call returns the same
bits as the eager call on the same input, for new input VALUES (copied into the graph's static buffer), in both tiers and both
families; a new prompt tensor, another input shape or an in-place weight edit is a new graph, never a stale one."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _s_pipe(wd):
    from omgsr_amd.diffusers_api import AutoencoderKL, UNet2DConditionModel
    from omgsr_amd.pipelines.omgsr_s import OMGSR_S_Infer
    from omgsr_amd.testing import seeded_init_
    vcfg = dict(block_out_channels=[32, 64, 128, 128], layers_per_block=1)
    ucfg = dict(block_out_channels=[64, 128, 256, 256], attention_head_dim=[1, 2, 4, 4], cross_attention_dim=128)
    v, u = seeded_init_(AutoencoderKL(**vcfg), 1, rounded=False), seeded_init_(UNet2DConditionModel(**ucfg), 2, rounded=False)
    return OMGSR_S_Infer(None, None, 273, DEV, wd, vae=v, unet=u)


@pytest.mark.parametrize("wd", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tiled_vae", [False, True])
def test_omgsr_s_graph_replay_equals_eager(wd, tiled_vae):
    from omgsr_amd import ops
    from omgsr_amd.testing import synthetic_lq
    try:
        pipe = _s_pipe(wd)
        if tiled_vae:
            pipe._init_tiled_vae(encoder_tile_size=128, decoder_tile_size=16)
        g = torch.Generator().manual_seed(5)
        prompt = torch.randn(1, 77, 128, generator=g).to(device=DEV, dtype=wd)
        pipe.vae.posterior_noise = torch.randn(2, 4, 24, 24, generator=g).to(DEV)
        xs = [synthetic_lq(2, 192, 192, seed=s).to(device=DEV, dtype=wd) for s in (1, 2, 3, 4)]
        with torch.no_grad():
            eager = [pipe(x, prompt, 16, 8)[0] for x in xs]              # latent 24 x 24 > 16 x 16: the tiled-latent path
            pipe.enable_graphs(True)
            got = [pipe(x, prompt, 16, 8)[0] for x in xs]                # eager, capture + replay, replay, replay
            assert pipe.graphs.captures == 1 and pipe.graphs.replays == 3
            for a, b in zip(eager, got):
                assert torch.equal(a, b)
            assert got[1].data_ptr() != got[2].data_ptr()                # fresh result tensors, like the reference's
            # another prompt TENSOR (new values): its K / V^T cache is rebuilt eagerly, then captured - never a stale replay
            prompt2 = (prompt * 0.5).contiguous()
            ref2 = None
            pipe.enable_graphs(False)
            ref2 = pipe(xs[0], prompt2, 16, 8)[0]
            pipe.enable_graphs(True)
            outs = [pipe(xs[0], prompt2, 16, 8)[0] for _ in range(3)]
            assert all(torch.equal(o, ref2) for o in outs) and not torch.equal(ref2, eager[0])
            # an in-place weight edit bumps the parameter version: new key, packed weights rebuilt, result follows the new weights
            pipe.unet.conv_out.weight.mul_(0.5)
            outs = [pipe(xs[0], prompt2, 16, 8)[0] for _ in range(3)]
            pipe.enable_graphs(False)
            ref3 = pipe(xs[0], prompt2, 16, 8)[0]
            assert all(torch.equal(o, ref3) for o in outs) and not torch.equal(ref3, ref2)
    finally:
        ops.set_compute_dtype(torch.bfloat16)


def test_graph_prompt_a_b_a_drops_the_stale_graph():
    """graph A bakes in the address of prompt A's cross-attention K / V^T, which the one-slot cache frees when prompt B
    replaces it. Returning to prompt A (same tensor object: the key hits graph A) must not replay against that freed memory: the graph is
    dropped (cache epoch), the call runs eagerly, the next one re-captures. Every result equals the eager result of its prompt."""
    from omgsr_amd import ops
    from omgsr_amd.testing import synthetic_lq
    wd = torch.bfloat16
    try:
        pipe = _s_pipe(wd)
        g = torch.Generator().manual_seed(7)
        pa = torch.randn(1, 77, 128, generator=g).to(device=DEV, dtype=wd)
        pb = torch.randn(1, 77, 128, generator=g).to(device=DEV, dtype=wd)
        pipe.vae.posterior_noise = torch.randn(1, 4, 24, 24, generator=g).to(DEV)
        x = synthetic_lq(1, 192, 192, seed=9).to(device=DEV, dtype=wd)
        with torch.no_grad():
            ref_a, ref_b = pipe(x, pa, 32, 8)[0], pipe(x, pb, 32, 8)[0]
            assert not torch.equal(ref_a, ref_b)
            pipe.enable_graphs(True)
            for _ in range(3):
                assert torch.equal(pipe(x, pa, 32, 8)[0], ref_a)           # eager, capture, replay
            assert pipe.graphs.captures == 1
            for _ in range(3):
                assert torch.equal(pipe(x, pb, 32, 8)[0], ref_b)           # rebuilds the K / V^T slot: graph A is now stale
            junk = [torch.full((1 << 20,), 7.0, device=DEV, dtype=wd) for _ in range(8)]      # re-use what the slot freed
            for _ in range(3):
                assert torch.equal(pipe(x, pa, 32, 8)[0], ref_a)           # dropped -> eager -> re-captured, never the stale replay
            assert pipe.graphs.stale_drops >= 1
            for _ in range(2):
                assert torch.equal(pipe(x, pb, 32, 8)[0], ref_b)
            del junk
            # a policy / guard / batch-invariance switch after a capture is a new key, not a stale replay
            ops.set_batch_invariant(True)
            try:
                inv = pipe(x, pa, 32, 8)[0]
                pipe.enable_graphs(False)
                assert torch.equal(pipe(x, pa, 32, 8)[0], inv)
            finally:
                ops.set_batch_invariant(False)
    finally:
        ops.set_compute_dtype(torch.bfloat16)


def test_graph_single_calls_a_b_a_never_capture_over_a_cold_cache():
    """prompt A, B, A as SINGLE calls. The third call has seen A once, so round 5 captured it - while the one-slot cross-attention
    K / V^T cache held B's: the builder ran under capture (recorded, never executed), the graph was discarded for the epoch change, and the eager
    re-run read the never-computed K / V^T. A capture now needs the preceding call to have had the same key with no rebuild since; every result of
    an alternating A / B service equals the eager result of its prompt, and consecutive calls still capture and replay."""
    from omgsr_amd import ops
    from omgsr_amd.testing import synthetic_lq
    wd = torch.bfloat16
    try:
        pipe = _s_pipe(wd)
        g = torch.Generator().manual_seed(11)
        pa = torch.randn(1, 77, 128, generator=g).to(device=DEV, dtype=wd)
        pb = torch.randn(1, 77, 128, generator=g).to(device=DEV, dtype=wd)
        pipe.vae.posterior_noise = torch.randn(1, 4, 24, 24, generator=g).to(DEV)
        x = synthetic_lq(1, 192, 192, seed=5).to(device=DEV, dtype=wd)
        with torch.no_grad():
            ref_a, ref_b = pipe(x, pa, 32, 8)[0], pipe(x, pb, 32, 8)[0]
            assert not torch.equal(ref_a, ref_b)
            pipe.enable_graphs(True)
            for i, (p, ref) in enumerate([(pa, ref_a), (pb, ref_b), (pa, ref_a), (pb, ref_b), (pa, ref_a), (pb, ref_b)]):
                junk = torch.full((1 << 22,), float("nan"), device=DEV, dtype=wd)       # poison what the allocator hands out next
                del junk
                assert torch.equal(pipe(x, p, 32, 8)[0], ref), f"alternating call {i}"
            assert pipe.graphs.captures == 0 and not pipe.graphs.capture_failures and not pipe.graphs._eager_only
            # the same pipe still graphs a prompt that stays: eager (re-warm), capture, replay
            for _ in range(3):
                assert torch.equal(pipe(x, pa, 32, 8)[0], ref_a)
            assert pipe.graphs.captures == 1 and pipe.graphs.replays >= 2
    finally:
        ops.set_compute_dtype(torch.bfloat16)


def test_omgsr_f_graph_batch_1_2_1():
    """The FLUX pipeline's cached timestep / guidance tensors are rebuilt per batch size: B = 1 -> 2 -> 1 under graphs equals eager."""
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer, prepare_latent_image_ids
    from omgsr_amd.testing import seeded_init_, synthetic_lq
    wd = torch.bfloat16
    try:
        vae = seeded_init_(AutoencoderKL(block_out_channels=[32, 64, 128, 128], layers_per_block=1, latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159), 3, rounded=False)
        flux = seeded_init_(FluxTransformer2DModel(num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=64,
                                                   pooled_projection_dim=32, in_channels=64), 4, rounded=False)
        pipe = OMGSR_F_Infer(None, None, DEV, wd, 244, 1.0, vae=vae, flux_transformer=flux)
        g = torch.Generator().manual_seed(6)
        pe, pooled = torch.randn(1, 32, 64, generator=g).to(DEV, wd), torch.randn(1, 32, generator=g).to(DEV, wd)
        tids, iids = torch.zeros(32, 3, device=DEV, dtype=wd), prepare_latent_image_ids(8, 8, DEV, wd)
        x1, x2 = synthetic_lq(1, 128, 128, seed=1).to(DEV, wd), synthetic_lq(2, 128, 128, seed=2).to(DEV, wd)
        n1, n2 = torch.randn(1, 16, 16, 16, generator=g).to(DEV), torch.randn(2, 16, 16, 16, generator=g).to(DEV)

        def call(x, n):
            pipe.vae.posterior_noise = n
            return pipe(x, pe, pooled, tids, iids, 16, 8)[0]
        with torch.no_grad():
            r1, r2 = call(x1, n1), call(x2, n2)
            pipe.enable_graphs(True)
            for x, n, r in [(x1, n1, r1)] * 3 + [(x2, n2, r2)] * 3 + [(x1, n1, r1)] * 3 + [(x2, n2, r2)] * 2:
                assert torch.equal(call(x, n), r)
        assert pipe.graphs.stale_drops >= 1
    finally:
        ops.set_compute_dtype(torch.bfloat16)


def test_omgsr_f_graph_replay_equals_eager():
    from omgsr_amd import ops
    from omgsr_amd.diffusers_api import AutoencoderKL, FluxTransformer2DModel
    from omgsr_amd.pipelines.omgsr_f import OMGSR_F_Infer, prepare_latent_image_ids
    from omgsr_amd.testing import seeded_init_, synthetic_lq
    wd = torch.float32
    try:
        vae = seeded_init_(AutoencoderKL(block_out_channels=[32, 64, 128, 128], layers_per_block=1, latent_channels=16, scaling_factor=0.3611, shift_factor=0.1159), 3, rounded=False)
        flux = seeded_init_(FluxTransformer2DModel(num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2, joint_attention_dim=64,
                                                   pooled_projection_dim=32, in_channels=64), 4, rounded=False)
        pipe = OMGSR_F_Infer(None, None, DEV, wd, 244, 1.0, vae=vae, flux_transformer=flux)
        g = torch.Generator().manual_seed(6)
        pe, pooled = torch.randn(1, 32, 64, generator=g).to(DEV), torch.randn(1, 32, generator=g).to(DEV)
        tids, iids = torch.zeros(32, 3, device=DEV), prepare_latent_image_ids(8, 8, DEV, wd)
        pipe.vae.posterior_noise = torch.randn(1, 16, 16, 16, generator=g).to(DEV)
        xs = [synthetic_lq(1, 128, 128, seed=s).to(DEV) for s in (1, 2, 3)]
        with torch.no_grad():
            eager = [pipe(x, pe, pooled, tids, iids, 16, 8)[0] for x in xs]
            pipe.enable_graphs(True)
            got = [pipe(x, pe, pooled, tids, iids, 16, 8)[0] for x in xs]
        assert pipe.graphs.captures == 1 and pipe.graphs.replays == 2
        for a, b in zip(eager, got):
            assert torch.equal(a, b)
    finally:
        ops.set_compute_dtype(torch.bfloat16)

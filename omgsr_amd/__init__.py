"""omgsr_amd — MI355X (gfx950) native implementation of OMGSR's single-mid-timestep SR inference
hot path (VAE encode -> one UNet / Flux forward at t* -> VAE decode) behind the diffusers module API.

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/omgsr_hip.h)
  _lib.py, ops.py  ctypes binding and tensor-level wrappers
  diffusers_api/   AutoencoderKL / UNet2DConditionModel / FluxTransformer2DModel / DDPMScheduler / PeftModel
                   with diffusers' names, configs, state-dict keys and forward() signatures
  pipelines/       OMGSR_S_Infer / OMGSR_F_Infer counterparts, latent tiling, tiled VAE (VAEHook)
  dist.py          one-process-per-GPU sharding + RCCL weight broadcast
"""
__version__ = "0.1.0"

"""Drop-in names for `from diffusers import ...` / `from peft import PeftModel` at the reference's
call sites (infer/omgsr_s_infer_model.py:3-4, infer/omgsr_f_infer_model.py:7-8)."""
from .autoencoder_kl import AutoencoderKL, FLUX_VAE_CONFIG, SD21_VAE_CONFIG  # noqa: F401
from .peft_compat import PeftModel  # noqa: F401
from .scheduling_ddpm import DDPMScheduler  # noqa: F401
from .transformer_flux import FLUX_DEV_CONFIG, FluxTransformer2DModel  # noqa: F401
from .unet_2d_condition import SD21_UNET_CONFIG, UNet2DConditionModel  # noqa: F401

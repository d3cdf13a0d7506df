"""dmhomo_amd — MI355X-native DGM denoising hot path of lhaippp/DMHomo.

    from dmhomo_amd.denoising_diffusion_models.classifier_free_guidance import Unet, GaussianDiffusion
    from dmhomo_amd.denoising_diffusion_models.denoising_diffusion_pytorch import Trainer

mirror the reference's import paths (DGM/dgm_sample.py:8-9).  Arithmetic runs in hand-written
gfx950 kernels (dmhomo_amd/csrc, C ABI in include/dmhomo_hip.h); PyTorch-ROCm carries memory,
streams and torch.distributed.  There is no CPU / eager fallback.
"""
from . import cfg, ddpm  # noqa: F401
from .cfg import Unet, GaussianDiffusion  # noqa: F401  (the conditional pair the DGM scripts use)
from .ddpm import Trainer  # noqa: F401

__all__ = ['cfg', 'ddpm', 'Unet', 'GaussianDiffusion', 'Trainer']

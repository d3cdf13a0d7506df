"""import-path alias of the reference module DGM/denoising_diffusion_models/classifier_free_guidance.py"""
from ..cfg import Unet, GaussianDiffusion, ModelPrediction  # noqa: F401
from ..ddpm import flow_warp  # noqa: F401

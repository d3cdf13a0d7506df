"""import-path alias of the reference module DGM/denoising_diffusion_models/classifier_free_guidance.py"""
from ..cfg import (Unet, GaussianDiffusion, ModelPrediction, extract, linear_beta_schedule,  # noqa: F401
                   cosine_beta_schedule)
from ..ddpm import flow_warp  # noqa: F401

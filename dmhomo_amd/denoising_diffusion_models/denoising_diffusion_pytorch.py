"""import-path alias of the reference module DGM/denoising_diffusion_models/denoising_diffusion_pytorch.py"""
from ..ddpm import (Unet, GaussianDiffusion, Trainer, adapt_homography_to_preprocessing_v3,  # noqa: F401
                    homo_to_flow, flow_warp, homo_gen, saveTrainPair, ModelPrediction, extract, linear_beta_schedule,
                    cosine_beta_schedule, mesh_grid, norm_grid, get_grid, DLT_solve, mesh_grid_np, get_flow_np)
from ..geometry import flow_to_image  # noqa: F401

"""CPU oracle for the DGM denoising hot path of lhaippp/DMHomo.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dmhomo_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may.  It is a plain-PyTorch/numpy (fp32 / f64, CPU) restatement of
the reference algorithm, written functionally over a ``state_dict`` so the same
weights drive the oracle and the HIP path.

Parity pin: the reference ships no tests and no golden vectors.
The oracle is pinned against outputs of the reference itself, captured in the
build container by ``tests/golden/make_golden.py`` (which imports
``/root/reference`` read-only) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every fixture on CPU.

Modules
-------
unet       N1-N11  conditional (CFG) and unconditional (DDP) UNet forward
diffusion  D1-D9   schedules, DDIM / DDPM samplers, q_sample, p_losses
geometry   G1-G7   homography rescale, homography->flow, flow->HSV image,
                   bilinear flow warp (explicit corner indices), DLT
"""
from . import unet, diffusion, geometry  # noqa: F401

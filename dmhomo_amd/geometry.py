"""numpy-facing wrapper of the flow -> HSV image kernel (reference signature DDP:1471-1486)."""
import numpy as np
import torch

from . import ops
from .ddpm import _dev


def flow_to_image(flow, max_flow=256):
    """G3: flow (H, W, 2) float32 numpy -> (H, W, 3) float32 RGB in [0, 1]."""
    f = np.asarray(flow, dtype=np.float32)
    if max_flow is not None:
        max_flow = max(max_flow, 1.)
    else:
        max_flow = float(np.max(f))
    t = torch.from_numpy(np.ascontiguousarray(f.transpose(2, 0, 1)))[None].to(_dev())
    rgb = ops.flow_to_image(t.contiguous(), max_flow)
    return rgb[0].permute(1, 2, 0).contiguous().cpu().numpy()

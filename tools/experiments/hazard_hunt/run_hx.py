"""round 2's open under-load case, rebuilt (`make -C dmhomo_amd/csrc hx`): the fully fused LinearAttention pass 2 with a
four-float LDS slot of 1.0 read after the exchange barrier and multiplied into the to_out scales.  Launches the block with
more workgroups than the chip holds and counts launches whose first row differs from the row computed alone.
    DMH_LIB_PATH=dmhomo_amd/libdmhomo_hx.so python tools/experiments/hazard_hunt/run_hx.py [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
C, H = 64, 128
g = (1 + 0.2 * rand((C,), 51)).to(dev)
pla = ops.PackedLinAttn(rand((384, C, 1, 1), 52, C ** -0.5).to(dev))
plo = ops.PackedLinAttnOut((rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0).to(dev), rand((C,), 54, 0.1).to(dev), (1 + 0.2 * rand((C,), 55)).to(dev))
for B in (25, 50):
    x = (rand((B, H, H, C), 50) * 1.3 + 0.2).to(dev)
    alone = ops.linear_attention_fused(x[:1].contiguous(), g, pla, 32 ** -0.5, out=plo)
    bad_launches, bad_pixels, worst = 0, 0, 0.0
    for it in range(N):
        y = ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)
        d = (y[:1] - alone).abs().amax(-1).reshape(-1)
        nb = int((d > 0).sum())
        bad_launches += int(nb > 0)
        bad_pixels += nb
        worst = max(worst, float(d.max()))
    print(f'{os.environ.get("DMH_LIB_PATH", "product")}: B={B}: {bad_launches} of {N} launches differ ({bad_pixels} pixels, max |diff| {worst:.3e})', flush=True)

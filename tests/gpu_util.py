"""helpers shared by the -m gpu parity tests (HIP path vs the CPU oracle)."""
import numpy as np
import torch


def dev():
    return torch.device('cuda', 0)


def nhwc(x):
    """NCHW cpu -> NHWC cuda contiguous"""
    return x.permute(0, 2, 3, 1).contiguous().to(dev())


def nchw(x):
    """NHWC cuda -> NCHW cpu"""
    return x.permute(0, 3, 1, 2).contiguous().cpu()


def report(name, got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    err = (got - ref).abs()
    denom = ref.abs().max().clamp_min(1e-30)
    print(f'[parity] {name}: max_abs={err.max().item():.3e} rel_to_max={(err.max() / denom).item():.3e} '
          f'ref_absmax={denom.item():.3e}')
    return err.max().item(), (err.max() / denom).item()


def close(name, got, ref, rtol, atol):
    report(name, got, ref)
    torch.testing.assert_close(got.cpu().to(ref.dtype), ref, rtol=rtol, atol=atol, msg=lambda m: f'{name}: {m}')


def close_rel(name, got, ref, rel):
    """max |got - ref| <= rel * max |ref|: the error measured against the tensor's own scale (what the kernels'
    fp32-accumulation error follows).  Tolerances in the GPU tests are set to <= 10x the error measured on MI355X."""
    a, r = report(name, got, ref)
    assert r <= rel, f'{name}: max error {a:.3e} = {r:.3e} of the reference scale, allowed {rel:.1e}'
    return r


class ReplayDeviceRng:
    """feeds recorded CPU draws to the product samplers (same interface as dmhomo_amd.cfg.DeviceRng)."""

    def __init__(self, draws):
        self.draws = [torch.as_tensor(np.asarray(d)) for d in draws]
        self.i = 0

    def _next(self, device):
        d = self.draws[self.i]
        self.i += 1
        return d.to(device)

    def randn(self, shape, device):
        d = self._next(device)
        assert tuple(d.shape) == tuple(shape)
        return d

    def uniform(self, n, device):
        d = self._next(device)
        assert d.shape == (n,)
        return d


def rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale

"""CPU: the public surface SURVEY 8b lists, name by name and signature by signature, against the list generated from the
imported reference (tests/golden/make_golden_r4.py -> tests/golden/surface.json; the reference itself never travels)."""
import importlib
import inspect
import json
import os

import pytest

# visualisation (a GIF dump through imageio): out of scope
OUT_OF_SCOPE = {('classifier_free_guidance', 'GaussianDiffusion', 'vis_bad_case')}


@pytest.fixture(scope='module')
def surface(golden_dir):
    with open(os.path.join(golden_dir, 'surface.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('modname', ['classifier_free_guidance', 'denoising_diffusion_pytorch'])
def test_alias_modules_offer_the_reference_surface(surface, modname):
    mod = importlib.import_module(f'dmhomo_amd.denoising_diffusion_models.{modname}')
    want = surface[modname]
    missing, different = [], []
    for name, sig in want['functions'].items():
        fn = getattr(mod, name, None)
        if fn is None:
            missing.append(name)
        elif str(inspect.signature(fn)) != sig:
            different.append((name, str(inspect.signature(fn)), sig))
    for cls_name, members in want.items():
        if cls_name == 'functions':
            continue
        cls = getattr(mod, cls_name)
        for name, sig in members.items():
            if (modname, cls_name, name) in OUT_OF_SCOPE:
                continue
            if not hasattr(cls, name):
                missing.append(f'{cls_name}.{name}')
            elif sig == 'property':
                if not isinstance(inspect.getattr_static(cls, name), property):
                    different.append((f'{cls_name}.{name}', 'not a property', sig))
            elif str(inspect.signature(getattr(cls, name))) != sig:
                different.append((f'{cls_name}.{name}', str(inspect.signature(getattr(cls, name))), sig))
    assert not missing, f'{modname}: missing {missing}'
    assert not different, f'{modname}: signatures differ {different}'


def test_host_side_helpers_match_the_reference_outputs(golden_dir):
    """the members that are host code in the reference too (integer grids, float64 schedules, extract)"""
    import numpy as np
    import torch
    from dmhomo_amd.denoising_diffusion_models import denoising_diffusion_pytorch as ddp
    gd = np.load(os.path.join(golden_dir, 'surface.npz'))
    mg = ddp.mesh_grid(2, 5, 7)
    assert mg.dtype == torch.int64 and np.array_equal(mg.numpy(), gd['mesh_grid'])
    mgn = ddp.mesh_grid_np(2, 6, 5)
    assert mgn.dtype == gd['mesh_grid_np'].dtype and np.array_equal(mgn, gd['mesh_grid_np'])
    sched = np.load(os.path.join(golden_dir, 'schedule.npz'))
    for name, fn in (('cosine', ddp.cosine_beta_schedule), ('linear', ddp.linear_beta_schedule)):
        b = fn(1000)
        assert b.dtype == torch.float64
        np.testing.assert_allclose(b.float().numpy(), sched[f'{name}1000.betas'], rtol=1e-6, atol=0)
    a = torch.arange(10.) * 2
    e = ddp.extract(a, torch.tensor([3, 9]), (2, 6, 4, 4))
    assert e.shape == (2, 1, 1, 1) and e.flatten().tolist() == [6., 18.]

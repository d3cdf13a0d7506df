"""-m gpu: de-duplication of the two classifier-free-guidance passes (cfg.Unet.dedup_dropped_rows) on device-side row
subsets (include/dmhomo_hip.h, "Row subsets").

The reference's conditional pass (CFG:404 -> CFG:415-425) replaces the class embedding of every row by the null embedding
with probability cond_drop_prob = 0.5 — also while sampling — so such a row has the inputs of its row in the null pass
(CFG:409) and the guided output ``null + (cond - null) * s`` (CFG:410) is the null output.  The product may skip those rows;
what these tests pin is that doing so changes NO bit of any result: per launch (the active rows of every tap of the trunk,
inactive rows of the result untouched), per guided forward, per sampling loop (eager and captured, both CFG schedules, torch's
generator and the sample-indexed one), at the headline configuration (bs = 25, s_step = 32, 128x128), and at the two
degenerate probabilities (0: nothing dropped, 1: everything dropped).
"""
import pytest
import torch

from gpu_util import dev, rand
from test_gpu_unet import make_cfg, g, _cond_inputs

pytestmark = pytest.mark.gpu


def test_rows_from_keep_list():
    """dmh_rows_from_keep: [n, kept rows ascending, B .. B + extra - 1]"""
    from dmhomo_amd import ops
    gen = torch.Generator().manual_seed(3)
    for B, extra in ((1, 0), (5, 0), (25, 0), (25, 25), (64, 3), (65, 0), (200, 200), (1000, 7)):
        for p in (0.0, 0.5, 1.0):
            keep = (torch.rand((B,), generator=gen) < p).to(torch.uint8)
            rows = ops.rows_from_keep(keep.to(dev()), extra=extra).cpu()
            want = [int(i) for i in keep.nonzero().flatten()] + list(range(B, B + extra))
            assert rows.shape == (1 + B + extra,) and rows.dtype == torch.int32
            assert int(rows[0]) == len(want) and rows[1:1 + len(want)].tolist() == want, (B, extra, p)


@pytest.mark.parametrize('dim,size,B', [(64, 32, 5), (64, 128, 3), (8, 16, 6), (16, 40, 4), (128, 32, 3)])
def test_trunk_row_subset_every_tap_bitwise(dim, size, B):
    """every launch of the trunk on a row subset: each tapped activation's ACTIVE rows are bit for bit those of the full
    launch (fp16-piece convs incl. the XCD re-deal over the active prefix, GroupNorm partials + finalize, the residual
    epilogues, fused / unfused LinearAttention, the bottleneck Attention, the fused final projection), for several subsets
    incl. the empty one and the full one; rows outside the subset of the caller's result buffer are not written"""
    from dmhomo_amd import ops
    m, _ = make_cfg(dim)
    x, rf, mk = _cond_inputs(B, size, 700 + dim)
    t = g(torch.full((B,), 499))
    c = g(torch.zeros(B, dtype=torch.long))
    keep_all = g(torch.ones(B, dtype=torch.uint8))
    x0 = m._stem(g(x), g(rf), g(mk))
    full_taps = {}
    full = m._run(None, t, c, None, None, [keep_all], taps=full_taps, x0=x0).clone()
    full_taps = {k: v.clone() for k, v in full_taps.items()}
    subsets = [[0], [B - 1], list(range(0, B, 2)), list(range(B)), []]
    for sub in subsets:
        keep = torch.zeros(B, dtype=torch.uint8)
        keep[sub] = 1
        rows = ops.rows_from_keep(g(keep))
        out = torch.full_like(full, 7.0)
        taps = {}
        # (the class embedding of every row is kept — keep_all — only the set of computed rows changes)
        got = m._run(None, t, c, None, None, [keep_all], taps=taps, x0=x0, out=out, rows=rows)
        off = [b for b in range(B) if b not in sub]
        if sub:
            assert torch.equal(got[sub], full[sub]), (dim, sub)
            for k, v in taps.items():
                if k == 'init_conv' or v is None:
                    continue
                assert torch.equal(v[sub], full_taps[k][sub]), (dim, sub, k)
        if off and dim <= 64 and ops.f16x3_default():   # (the fused final projection writes the caller's buffer; wider models
                                                        #  and the exact-fp32 conv variants copy all rows into it)
            assert bool((out[off] == 7.0).all()), (dim, sub)          # the launch never wrote the inactive rows


@pytest.mark.parametrize('dim,size,B', [(64, 128, 5), (8, 32, 7)])
def test_forward_with_cond_scale_dedup_bitwise(dim, size, B):
    """forward_with_cond_scale (CFG:403-410) with dedup_dropped_rows: bit for bit the full 2B-row result in both CFG
    schedules, for several class-dropout draws"""
    m, _ = make_cfg(dim)
    x, rf, mk = _cond_inputs(B, size, 520)
    t = g(torch.full((B,), 967))
    c = g(torch.zeros(B, dtype=torch.long))
    for seed in (5, 6, 7):
        outs = {}
        for mode in ('batched', 'streams'):
            for dd in (False, True):
                m.cfg_mode, m.dedup_dropped_rows = mode, dd
                torch.manual_seed(seed)
                outs[mode, dd] = m.forward_with_cond_scale(g(x), t, c, g(rf), g(mk), cond_scale=3.).clone()
        ref = outs['batched', False]
        assert torch.isfinite(ref).all()
        for k, v in outs.items():
            assert torch.equal(v, ref), (seed, k)
    m.cfg_mode, m.dedup_dropped_rows = 'batched', False


@pytest.mark.parametrize('p', [0.0, 1.0, 0.3])
def test_dedup_at_other_drop_probabilities(p):
    """cond_drop_prob = 0: no mask is drawn and nothing is skipped; 1: every conditional row is the null row (the active list
    is empty: every workgroup of the conditional pass retires); 0.3: fewer drops"""
    from dmhomo_amd import cfg
    m, _ = make_cfg(8, cond_drop_prob=p)
    x, rf, mk = _cond_inputs(4, 32, 530)
    t, c = g(torch.full((4,), 500)), g(torch.zeros(4, dtype=torch.long))
    for mode in ('batched', 'streams'):
        m.cfg_mode = mode
        outs = []
        for dd in (False, True):
            m.dedup_dropped_rows = dd
            torch.manual_seed(1)
            outs.append(m.forward_with_cond_scale(g(x), t, c, g(rf), g(mk), cond_scale=3.).clone())
        assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all(), (p, mode)


def test_sampler_step_keep_takes_the_null_row():
    """dmh_sampler_step(keep=): a row with keep == 0 never reads model_cond (NaN there must not reach the result) and comes
    out as the null row's guided value; kept rows are unchanged"""
    from dmhomo_amd import ops
    from dmhomo_amd._lib import DmhStep
    B = 4
    cond, null, x, nz = (g(rand((B, 6, 8, 8), 40 + i)) for i in range(4))
    keep = g(torch.tensor([1, 0, 1, 0], dtype=torch.uint8))
    step = DmhStep(objective=1, clip=1, mode=ops.MODE_DDIM, cond_scale=3., sqrt_recip_ac=1.5, sqrt_recipm1_ac=1.1,
                   sqrt_ac=0.7, sqrt_1m_ac=0.7, c0=0.9, c1=0.3, c2=0.2)
    ref_in = cond.clone()
    ref_in[[1, 3]] = null[[1, 3]]
    want = ops.sampler_step(step, ref_in, null, x, nz, True, True)
    poisoned = cond.clone()
    poisoned[[1, 3]] = float('nan')
    got = ops.sampler_step(step, poisoned, null, x, nz, True, True, keep=keep)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize('mode', ['batched', 'streams'])
@pytest.mark.parametrize('keyed', [False, True])
def test_sample_dedup_eager_and_graph_bitwise(mode, keyed):
    """the sampling loop with dedup_dropped_rows, eager and as the captured step (the row list is device data inside the
    graph: one graph for every step, no host read): bitwise the full 2B-row loop, with torch's generator and with the
    sample-indexed one, on the capturing call and on replays with other inputs / draws"""
    from dmhomo_amd import cfg
    m, _ = make_cfg(8)
    m.cfg_mode = mode
    S, size, B = 6, 32, 5
    d = cfg.GaussianDiffusion(m, image_size=size, timesteps=1000, sampling_timesteps=S, objective='pred_x0').to(dev())
    _, rf, mk = _cond_inputs(B, size, 900)
    rf01, flow, c = g((rf + 1) / 2), g(rand((B, 2, size, size), 903)), g(torch.zeros(B, dtype=torch.long))
    mk = g(mk)

    def run(graph, dedup, seed, rf_in):
        d.hip_graph, m.dedup_dropped_rows = graph, dedup
        if keyed:
            d.rng.key_by_sample(seed, range(40, 40 + B), dev())
        else:
            torch.manual_seed(seed)
        return d.sample(c, rf_in, flow, mk)[0].clone()
    for seed, rf_in in ((5, rf01), (6, 1 - rf01), (5, rf01)):
        ref = run(False, False, seed, rf_in)
        assert torch.isfinite(ref).all()
        assert torch.equal(run(False, True, seed, rf_in), ref), (seed, 'eager dedup')
        assert torch.equal(run(True, True, seed, rf_in), ref), (seed, 'graph dedup')
        assert torch.equal(run(True, False, seed, rf_in), ref), (seed, 'graph full')
    d.hip_graph, m.dedup_dropped_rows, m.cfg_mode = False, False, 'batched'
    d.rng.unkey()


def test_sample_bs25_s32_dedup_graph_bitwise():
    """configs[1] (dim 64, 128x128, bs = 25, s_step = 32, 'streams', captured step, noise keyed by sample id) with
    dedup_dropped_rows: bit for bit the full 2B-row samples; and in 'batched' mode too"""
    from dmhomo_amd import cfg
    from test_gpu_rng import _fullsize_inputs
    m, _ = make_cfg(64)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    d.hip_graph = True
    rf01, flow, mk, c = (t.to(dev()) for t in _fullsize_inputs(25))
    outs = {}
    for mode, dd in (('streams', False), ('streams', True), ('batched', True)):
        m.cfg_mode, m.dedup_dropped_rows = mode, dd
        d.rng.key_by_sample(7, range(25), dev())
        outs[mode, dd] = d.sample(c, rf01, flow, mk)[0].clone()
    m.cfg_mode, m.dedup_dropped_rows = 'batched', False
    ref = outs['streams', False]
    assert torch.isfinite(ref).all() and float(ref.min()) >= 0 and float(ref.max()) <= 1
    assert torch.equal(outs['streams', True], ref)
    assert torch.equal(outs['batched', True], ref)


@pytest.mark.parametrize('graph', [False, True])
def test_invalid_class_id_poisons_its_row_only(graph):
    """a class id outside the embedding table (the reference's nn.Embedding raises, CFG:419) is device data here: the row comes
    out NaN — in the eager path (dmh_class_embed) and in the replayed step (dmh_ss_gather) — and the other rows are untouched"""
    from dmhomo_amd import cfg
    m, _ = make_cfg(8, cond_drop_prob=0.)            # every class kept: the bad id is always looked up
    m.cfg_mode = 'streams'
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=3, objective='pred_x0').to(dev())
    d.hip_graph = graph
    _, rf, mk = _cond_inputs(3, 16, 910)
    rf01, flow, mk = g((rf + 1) / 2), g(rand((3, 2, 16, 16), 911)), g(mk)
    outs = []
    for classes in ([0, 0, 0], [0, 5, 0], [0, -1, 0]):
        d.rng.key_by_sample(3, range(3), dev())
        outs.append(d.sample(g(torch.tensor(classes)), rf01, flow, mk)[0].clone())
    d.hip_graph = False
    d.rng.unkey()
    assert torch.isfinite(outs[0]).all()
    for bad in outs[1:]:
        assert torch.isnan(bad[1]).all()
        assert torch.equal(bad[[0, 2]], outs[0][[0, 2]])


@pytest.mark.parametrize('variant', ['0', '6'])
def test_row_subsets_under_exact_fp32_conv_variants_in_child_process(variant):
    """DMH_CONV3_VARIANT = 0 (exact-fp32 implicit GEMM, conv.hip) / 6 (exact-fp32 Winograd, conv_wino.hip) also take DmhConv.rows:
    the variant is read once per process, so the subset tests run in a fresh child under it"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMH_CONV3_VARIANT=variant)
    me = os.path.join(root, 'tests', 'test_gpu_dedup.py')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        me + '::test_trunk_row_subset_every_tap_bitwise', me + '::test_forward_with_cond_scale_dedup_bitwise',
                        me + '::test_sample_dedup_eager_and_graph_bitwise'],
                       capture_output=True, text=True, env=env, cwd=root, timeout=1500)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]

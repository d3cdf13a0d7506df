"""-m gpu: the sample-indexed noise generator (dmh_rng_indexed, csrc/rng.hip) against its numpy restatement
(oracle/rng.py, pinned to Random123's Philox4x32-10 known-answer vectors in tests/test_oracle_golden.py), and what
SURVEY 8e asks of it: an N-rank sharded sampling run equals the single-process run row for row, bit for bit —
on the FAST path (per-step HIP graph on, 'streams' CFG mode).

Bars: the integer stream (raw Philox words, uniform draws) bit-exact; the fp32 Box-Muller normals within 2e-6 abs of the
numpy evaluation (fp32 log / sincospi of two libraries); sharded vs whole batch: torch.equal."""
import numpy as np
import pytest
import torch

from gpu_util import dev
from oracle import rng as ORNG

pytestmark = pytest.mark.gpu


def _state(seed, draw=0):
    return torch.tensor([seed, draw, 0, 0], dtype=torch.int64, device=dev())


@pytest.mark.parametrize('per', [1, 3, 4, 6, 1021, 6 * 16 * 16])
def test_words_and_uniform_bit_exact(per):
    from dmhomo_amd import ops
    ids_h = [0, 1, 7, 2 ** 32 + 5, 2 ** 40 + 123456789, 24]
    ids = torch.tensor(ids_h, dtype=torch.int64, device=dev())
    seed = 0x1234_5678_9ABC_DEF
    for draw in (0, 1, 77):
        st = _state(seed, draw)
        w = ops.rng_indexed((len(ids_h), per), ids, st, 2).cpu().numpy().view(np.uint32)
        assert np.array_equal(w, ORNG.words(seed, ids_h, draw, per))
        assert st.cpu().tolist() == [seed, draw + 1, 0, 0]           # the launch advanced the draw index, tickets back at 0
        u = ops.rng_indexed((len(ids_h), per), ids, st, 1).cpu().numpy()
        assert np.array_equal(u, ORNG.uniform(seed, ids_h, draw + 1, per))
        assert u.min() >= 0 and u.max() < 1


def test_normals_vs_oracle_and_distribution():
    from dmhomo_amd import ops
    from scipy import stats
    ids_h = list(range(100, 125))
    ids = torch.tensor(ids_h, dtype=torch.int64, device=dev())
    st = _state(99, 5)
    z = ops.rng_indexed((25, 6, 128, 128), ids, st, 0)
    ref = ORNG.randn(99, ids_h, 5, (6, 128, 128))
    err = float(np.abs(z.cpu().numpy() - ref).max())
    print(f'[parity] dmh_rng_indexed normals vs numpy Box-Muller: max_abs={err:.3e}')
    assert err <= 2e-6
    zz = z.double().flatten()
    assert abs(float(zz.mean())) < 3e-3 and abs(float(zz.std()) - 1) < 3e-3 and float(zz.abs().max()) < 6.8
    assert abs(float((zz ** 3).mean())) < 1e-2 and abs(float((zz ** 4).mean()) - 3) < 3e-2
    assert stats.kstest(zz[:400000].cpu().numpy(), 'norm').pvalue > 1e-3
    # rows of successive draws and of neighbouring samples are uncorrelated
    z2 = ops.rng_indexed((25, 6, 128, 128), ids, st, 0).double().flatten()
    assert abs(float((zz * z2).mean())) < 3e-3
    a, b = z[3].double().flatten(), z[4].double().flatten()
    assert abs(float((a * b).mean())) < 2e-2


def test_rows_do_not_depend_on_batch_or_position():
    from dmhomo_amd import ops
    full_ids = torch.arange(40, 52, dtype=torch.int64, device=dev())
    full = ops.rng_indexed((12, 6, 32, 32), full_ids, _state(3, 9), 0)
    pick = [49, 41, 51]
    part = ops.rng_indexed((3, 6, 32, 32), torch.tensor(pick, dtype=torch.int64, device=dev()), _state(3, 9), 0)
    assert torch.equal(part, full[[p - 40 for p in pick]])
    other = ops.rng_indexed((3, 6, 32, 32), torch.tensor(pick, dtype=torch.int64, device=dev()), _state(4, 9), 0)
    assert not torch.equal(other, part)


def test_captured_launch_replays_successive_draws():
    """the draw index is device state advanced by the launch itself: a HIP graph holding one draw yields draw k, k+1, ..."""
    from dmhomo_amd import cfg
    rng = cfg.DeviceRng().key_by_sample(21, range(8, 12), dev())
    shape = (4, 6, 16, 16)
    eager = [rng.randn(shape, dev()).clone() for _ in range(4)]
    rng.key_by_sample(21, range(8, 12), dev())                 # same storage, draw index back at 0
    side = torch.cuda.Stream(device=dev())
    snap = rng.snapshot(dev())
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        rng.randn(shape, dev())
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode='thread_local'):
        out = rng.randn(shape, dev())
    rng.restore(snap, dev())
    for k in range(4):
        gr.replay()
        assert torch.equal(out, eager[k]), k
    assert rng.state.cpu().tolist() == [21, 4, 0, 0]


def _fullsize_inputs(B, size=128):
    g = torch.Generator().manual_seed(4242)
    rf01 = torch.rand((B, 3, size, size), generator=g)
    mk = (torch.rand((B, 1, size, size), generator=g) > 0.4).float()
    flow = torch.randn((B, 2, size, size), generator=g)
    return rf01, flow, mk, torch.zeros(B, dtype=torch.long)


def test_sharded_graph_sampling_equals_whole_batch():
    """configs[1] (dim 64, 128x128, bs = 25, s_step = 32) with hip_graph = True and 'streams': the whole batch equals,
    BITWISE, the concatenation of two 'virtual ranks' holding rows 0-12 and 13-24 (each keyed with its own slice of the
    global sample ids, as distributed.key_noise_by_sample does per rank), and rows 0-1 equal a bs = 2 run — the N-GPU
    output is the concatenation of the shards whatever N is.  A second call continues the draw index (fresh noise)."""
    from dmhomo_amd import cfg
    from dmhomo_amd import distributed as D
    from test_gpu_unet import make_cfg
    m, _ = make_cfg(64)
    m.cfg_mode = 'streams'
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    d.hip_graph = True
    ins = [t.to(dev()) for t in _fullsize_inputs(25)]

    def run(lo, hi, world_rank=None):
        if world_rank is None:
            d.rng.key_by_sample(7, range(lo, hi), dev())
        else:                                                       # the key a rank of an N-rank job derives
            seed, ids = D.noise_key(7, 25, world_rank[1], world_rank[0])
            assert (ids.start, ids.stop) == (lo, hi)
            d.rng.key_by_sample(seed, ids, dev())
        rf01, flow, mk, c = (t[lo:hi].contiguous() for t in ins)
        img, _, _ = d.sample(c, rf01, flow, mk)
        return img

    whole = run(0, 25)
    assert d.__dict__.get('_graph_state') is not None              # the keyed generator stays on the graph path
    assert torch.isfinite(whole).all() and float(whole.min()) >= 0 and float(whole.max()) <= 1
    shard0, shard1 = run(0, 13, (2, 0)), run(13, 25, (2, 1))
    assert torch.equal(torch.cat([shard0, shard1]), whole)
    assert torch.equal(run(0, 2), whole[:2])
    assert not torch.equal(whole[0], whole[1])                       # rows draw different noise
    # the draw index keeps counting across calls: a second batch under the same key gets new noise, the same on any N
    d.rng.key_by_sample(7, range(0, 25), dev())
    rf01, flow, mk, c = ins
    first, _, _ = d.sample(c, rf01, flow, mk)
    second, _, _ = d.sample(c, rf01, flow, mk)
    assert torch.equal(first, whole) and not torch.equal(second, first)
    d.rng.key_by_sample(7, range(13, 25), dev())
    for _ in range(2):
        tail, _, _ = d.sample(*(t[13:].contiguous() for t in (c, rf01, flow, mk)))
    assert torch.equal(tail, second[13:])
    m.cfg_mode = 'batched'


def test_keyed_graph_equals_keyed_eager():
    """the keyed generator inside the captured denoise step draws what the eager loop draws (same kernel, the draw index
    is device state either way) — on the capturing call too (the warm-up's draws are put back)."""
    from dmhomo_amd import cfg
    from test_gpu_unet import make_cfg
    m, _ = make_cfg(8)
    for mode in ('batched', 'streams'):
        m.cfg_mode = mode
        d = cfg.GaussianDiffusion(m, image_size=32, timesteps=1000, sampling_timesteps=6, objective='pred_x0').to(dev())
        rf01, flow, mk, c = (t.to(dev()) for t in _fullsize_inputs(3, 32))
        outs = {}
        for graph in (False, True, True):
            d.hip_graph = graph
            d.rng.key_by_sample(5, range(30, 33), dev())
            outs.setdefault(graph, []).append(d.sample(c, rf01, flow, mk)[0].clone())
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][0])
    m.cfg_mode = 'batched'


def test_keep_mask_is_the_uniform_draw_compared_on_the_device():
    """dmh_rng_keep_mask (the class-dropout mask of CFG:84-90 in one launch) == (dmh_rng_indexed kind 1 < prob) for the same
    key and draw, and advances the draw index by one"""
    from dmhomo_amd import ops
    ids_h = list(range(500, 525)) + [2 ** 33 + 7]
    ids = torch.tensor(ids_h, dtype=torch.int64, device=dev())
    for prob in (0.5, 0.1, 0.9):
        st = _state(1234, 9)
        keep = ops.rng_keep_mask(ids, st, prob)
        assert keep.dtype == torch.uint8 and st.cpu().tolist() == [1234, 10, 0, 0]
        want = (ORNG.uniform(1234, ids_h, 9)[:, 0] < np.float32(prob)).astype(np.uint8)
        assert np.array_equal(keep.cpu().numpy(), want)
    assert 0 < int(ops.rng_keep_mask(torch.arange(4096, dtype=torch.int64, device=dev()), _state(5), 0.5).sum()) < 4096


def test_short_last_batch_draws_the_first_ids():
    """a loader that keeps its short last batch (dataset.ConditionLoader, like the reference's DataLoader, DDP:1746-1752) hands
    sample() fewer rows than the generator was keyed for: the draw uses the first n ids (cfg.DeviceRng.ids_for) — bitwise what
    a generator keyed for exactly those ids draws at the same draw index — also on the captured path; more rows than ids raise"""
    from dmhomo_amd import cfg
    from test_gpu_unet import make_cfg
    m, _ = make_cfg(8)
    m.cfg_mode = 'streams'
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective='pred_x0').to(dev())
    rf01, flow, mk, c = (t.to(dev()) for t in _fullsize_inputs(3, 16))
    for graph in (False, True):
        d.hip_graph = graph
        d.rng.key_by_sample(9, range(10, 13), dev())
        d.sample(c, rf01, flow, mk)
        short = d.sample(c[:2], rf01[:2].contiguous(), flow[:2].contiguous(), mk[:2].contiguous())[0].clone()
        d.rng.key_by_sample(9, range(10, 12), dev())
        d.sample(c[:2], rf01[:2].contiguous(), flow[:2].contiguous(), mk[:2].contiguous())       # (same number of draws as above)
        want = d.sample(c[:2], rf01[:2].contiguous(), flow[:2].contiguous(), mk[:2].contiguous())[0]
        assert torch.equal(short, want), graph
    with pytest.raises(ValueError):
        d.rng.key_by_sample(9, range(2), dev())
        d.sample(c, rf01, flow, mk)
    d.hip_graph = False
    d.rng.unkey()
    m.cfg_mode = 'batched'

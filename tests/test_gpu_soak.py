"""-m gpu soak: timing-dependent faults (a register read before an MFMA wrote it, an LDS slot reused a barrier early) do not
show up in a parity case that launches a kernel once on an idle chip — they show up as a few differing pixels in SOME
launches when the chip holds two workgroups per CU and other kernels run beside them (it happened once to the fused
LinearAttention: DESIGN.md 3.1, commit 25ebbcb).  Each case launches a kernel with more workgroups than the chip holds,
many times, alone and with a second stream keeping the CUs busy, and compares the first rows bitwise with the same rows
computed alone.  (Was tools/soak_rows.py; the whole file runs in well under a minute.)"""
import pytest
import torch

from gpu_util import dev, rand

pytestmark = pytest.mark.gpu
LAUNCHES = 60


@pytest.fixture(scope='module')
def ops():
    from dmhomo_amd import ops as O
    return O


class _Neighbour:
    """keeps a second stream busy with launches of ANOTHER kernel family while the soaked launches run, so that workgroups
    of different kernels share CUs the way they do in the two-stream sampling step"""

    def __init__(self, ops, kind):
        self.ops, self.kind = ops, kind
        self.stream = torch.cuda.Stream(device=dev())
        if kind == 'conv':
            w = rand((128, 128, 3, 3), 90, (1.0 / 1152) ** 0.5).to(dev())
            self.pc = ops.PackedConv(w, None, 128)
            self.x = rand((25, 64, 64, 128), 91).to(dev())
        elif kind == 'linattn':
            self.g = (1 + 0.2 * rand((128,), 92)).to(dev())
            self.pla = ops.PackedLinAttn(rand((384, 128, 1, 1), 93, 128 ** -0.5).to(dev()))
            self.x = rand((25, 64, 64, 128), 94).to(dev())

    def kick(self, n=2):
        if self.kind is None:
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                if self.kind == 'conv':
                    self.ops.conv2d(self.pc, self.x)
                else:
                    self.ops.linear_attention_fused(self.x, self.g, self.pla, 32 ** -0.5)

    def join(self):
        if self.kind is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


@pytest.mark.parametrize('neighbour', [None, 'conv'])
@pytest.mark.parametrize('B,H', [(25, 128), (50, 128), (50, 64)])
def test_soak_fused_linear_attention_block(ops, B, H, neighbour):
    """Residual(PreNorm(LinearAttention)) of the 64-channel levels (pixel stats + pass 1 + merge + fully fused pass 2):
    rows 0-1 of a B-row launch, 60 launches, bitwise equal to the rows computed alone"""
    C = 64
    gq = (1 + 0.2 * rand((C,), 51)).to(dev())
    pla = ops.PackedLinAttn(rand((384, C, 1, 1), 52, C ** -0.5).to(dev()))
    plo = ops.PackedLinAttnOut((rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0).to(dev()), rand((C,), 54, 0.1).to(dev()),
                               (1 + 0.2 * rand((C,), 55)).to(dev()))
    x = (rand((B, H, H, C), 50) * 1.3 + 0.2).to(dev())
    alone = ops.linear_attention_fused(x[:2].contiguous(), gq, pla, 32 ** -0.5, out=plo)
    nb = _Neighbour(ops, neighbour)
    bad = 0
    for _ in range(LAUNCHES):
        nb.kick()
        y = ops.linear_attention_fused(x, gq, pla, 32 ** -0.5, out=plo)
        bad += int(not torch.equal(y[:2], alone))
    nb.join()
    assert bad == 0, f'{bad} of {LAUNCHES} launches differ from the rows computed alone'


@pytest.mark.parametrize('neighbour', [None, 'linattn'])
@pytest.mark.parametrize('C,H', [(128, 64), (256, 32)])
def test_soak_fused_linear_attention_core(ops, C, H, neighbour):
    """the two fused passes without the fused to_out (the deeper levels), 50 rows"""
    B = 50
    gq = (1 + 0.2 * rand((C,), 61)).to(dev())
    pla = ops.PackedLinAttn(rand((384, C, 1, 1), 62, C ** -0.5).to(dev()))
    x = (rand((B, H, H, C), 60) * 1.1 - 0.1).to(dev())
    alone = ops.linear_attention_fused(x[:2].contiguous(), gq, pla, 32 ** -0.5)
    nb = _Neighbour(ops, neighbour)
    bad = 0
    for _ in range(LAUNCHES):
        nb.kick()
        y = ops.linear_attention_fused(x, gq, pla, 32 ** -0.5)
        bad += int(not torch.equal(y[:2], alone))
    nb.join()
    assert bad == 0, f'{bad} of {LAUNCHES} launches differ from the rows computed alone'


@pytest.mark.parametrize('neighbour', [None, 'linattn'])
@pytest.mark.parametrize('B', [25, 50])
def test_soak_canonical_conv(ops, B, neighbour):
    """the canonical fused conv3x3 + GN + SiLU launch (64 -> 64 @ 128x128, prologue, GroupNorm partials): outputs and
    partials of rows 0-1, 60 launches, bitwise equal to the rows computed alone"""
    w = rand((64, 64, 3, 3), 1, (1.0 / 576) ** 0.5).to(dev())
    pc = ops.PackedConv(w, rand((64,), 2, 0.1).to(dev()), 64)
    x0 = rand((B, 128, 128, 64), 3).to(dev())
    coef = torch.stack([1 + 0.1 * rand((B, 64), 5), 0.1 * rand((B, 64), 6)], 1).contiguous().to(dev())
    a, sa = ops.conv2d(pc, x0[:2].contiguous(), in_coef=coef[:2].contiguous(), want_stats=True)
    nb = _Neighbour(ops, neighbour)
    bad = 0
    for _ in range(LAUNCHES):
        nb.kick()
        b, sb = ops.conv2d(pc, x0, in_coef=coef, want_stats=True)
        bad += int(not (torch.equal(b[:2], a) and torch.equal(sb[:2], sa)))
    nb.join()
    assert bad == 0, f'{bad} of {LAUNCHES} launches differ from the rows computed alone'


def test_soak_deep_conv_two_streams(ops):
    """512 -> 512 @ 16x16 (the 2x2-wave workgroup layout) on two streams at once, 25 rows each: the streams' results are
    bitwise those of the same launch alone"""
    w = rand((512, 512, 3, 3), 11, (1.0 / 4608) ** 0.5).to(dev())
    pc = ops.PackedConv(w, None, 512)
    x = rand((25, 16, 16, 512), 12).to(dev())
    alone = ops.conv2d(pc, x)
    s1, s2 = torch.cuda.Stream(device=dev()), torch.cuda.Stream(device=dev())
    bad = 0
    for _ in range(LAUNCHES // 2):
        outs = []
        for s in (s1, s2):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                outs.append(ops.conv2d(pc, x))
        for s in (s1, s2):
            torch.cuda.current_stream().wait_stream(s)
        bad += sum(int(not torch.equal(o, alone)) for o in outs)
    assert bad == 0, f'{bad} launches differ'


@pytest.mark.parametrize('B', [25, 50])
def test_soak_ring_kernel_beside_a_conv_240_launches(ops, B):
    """pass 1 of the fused LinearAttention at C = 64 / 128x128 — linattn_kv_ring_kernel, the LDS-DMA ring with raw barriers,
    a sibling of round 3's software-pipelined form that was bitwise right alone and wrong in 58 of 60 launches beside a
    convolution on a second stream (unexplained; tools/experiments/README.md) — on its own: 240 launches with a conv
    running beside it, EVERY word of the per-split (max, sum, context) partials of all rows bitwise equal to the launch made
    on an idle chip.  (The build-time counterpart: tests/test_isa_hazards.py checks that every barrier of the kernel is
    reached with none of the wave's DMA pieces in flight.)"""
    from dmhomo_amd.ops import _empty, ptr, call, lib
    C, H = 64, 128
    n = H * H
    gq = (1 + 0.2 * rand((C,), 61)).to(dev())
    pla = ops.PackedLinAttn(rand((384, C, 1, 1), 62, C ** -0.5).to(dev()))
    x = (rand((B, H, H, C), 63) * 1.3 + 0.2).to(dev())
    stats = _empty((B, n, 2), x)
    call('dmh_pixel_stats', ptr(x), ptr(stats), B * n, C, 1e-5, None, 0)
    ns = lib().dmh_linattn_fused_splits(B, n)

    def context():
        partial = torch.zeros((B, ns, 4, 1088), device=dev())
        call('dmh_linattn_fused_context', ptr(x), ptr(stats), ptr(gq), ptr(pla.wpack), ptr(partial), B, n, C, None)
        return partial
    torch.cuda.synchronize()
    alone = context()
    torch.cuda.synchronize()
    nb = _Neighbour(ops, 'conv')
    bad = 0
    for _ in range(240):
        nb.kick()
        bad += int(not torch.equal(context(), alone))
    nb.join()
    assert bad == 0, f'{bad} of 240 launches differ from the launch on an idle chip'


def test_soak_whole_step_is_reproducible_call_after_call():
    """the whole headline step as it is benchmarked — dim 64, 128x128, bs 25, s_step 32, the two CFG passes concurrent on two
    streams, the per-step graph replayed, noise keyed by sample — 16 calls under the same key: every call BITWISE the first
    (round 4 ran 600 such calls, 6.3 M kernel launches: 0 differed; tools/experiments/step_soak.py)"""
    from dmhomo_amd import cfg, ddpm
    torch.manual_seed(0)
    model = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    model.cfg_mode = 'streams'
    d = cfg.GaussianDiffusion(model, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    d.hip_graph = True
    data, classes = next(ddpm.SyntheticConditions(128, 25, seed=1000, device=dev()))
    rgb_flow, flow, mask = data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous()

    def run():
        d.rng.key_by_sample(99, range(25), dev())
        return d.sample(classes, rgb_flow, flow, mask)[0].clone()
    first = run()
    assert torch.isfinite(first).all()
    bad = sum(int(not torch.equal(run(), first)) for _ in range(16))
    assert bad == 0, f'{bad} of 16 calls differ from the first'

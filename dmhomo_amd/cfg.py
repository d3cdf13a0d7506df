"""Conditional UNet + GaussianDiffusion with classifier-free guidance on MI355X.

Host-side mirror of the reference's
``DGM/denoising_diffusion_models/classifier_free_guidance.py`` (tag CFG): same class
names, constructor signatures, method names, ``state_dict`` keys and RNG call order,
so scripts written against the reference (``DGM/dgm_sample.py:28-38``) run
unchanged — but every tensor value is produced by the gfx950 kernels of
libdmhomo_hip.so (see ``engine.py``).  There is no CPU path.
"""
from collections import OrderedDict, namedtuple

import torch
from torch import nn

from . import _params as P
from . import ops
from ._lib import DmhStep
from .engine import UnetEngine
from .schedule import make_buffers, ddim_pairs, linear_beta_schedule, cosine_beta_schedule  # noqa: F401

ModelPrediction = namedtuple('ModelPrediction', ['pred_noise', 'pred_x_start'])


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def extract(a, t, x_shape):
    """D3, CFG:472-475: a.gather(-1, t) shaped (b, 1, ..., 1) (integer indexing: tensor plumbing)."""
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


class DeviceRng:
    """The noise source of the samplers, drawn in the reference's call order (CFG:679 randn(shape), CFG:90
    zeros(B).uniform_(0,1), CFG:703 randn_like).

    Default: torch's generator of the target device — the reference's own behaviour: row b of a draw depends on how many
    rows the process holds and on the process's seed.
    After ``key_by_sample(seed, sample_ids)``: every value is a pure function of (seed, GLOBAL sample id, draw index,
    element) (``dmh_rng_indexed``: Philox4x32-10 + Box-Muller; the draw index lives in device memory and advances with
    every launch, also inside a replayed HIP graph) — a batch sharded over N ranks, each keyed with its own slice of the
    sample ids, reproduces the single-GPU batch row for row, bit for bit."""

    def __init__(self):
        self.sample_ids = None           # (B,) int64 device tensor once keyed
        self.state = None                # (4,) int64 device tensor: seed, draw index, tickets, reserved

    def key_by_sample(self, seed, sample_ids, device=None):
        """sample_ids: the GLOBAL indices of this process's rows (``range(lo, hi)`` of distributed.shard_bounds), in row
        order.  Re-keying with the same number of rows on the same device keeps the tensors' storage (a captured graph
        stays valid) and restarts the draw index at 0."""
        ids = torch.as_tensor(list(sample_ids), dtype=torch.int64)
        device = torch.device(device) if device is not None else (self.sample_ids.device if self.sample_ids is not None
                                                                 else torch.device('cuda', torch.cuda.current_device()))
        state = torch.tensor([int(seed), 0, 0, 0], dtype=torch.int64)
        if self.sample_ids is not None and self.sample_ids.shape == ids.shape and self.sample_ids.device == device:
            self.sample_ids.copy_(ids)
            self.state.copy_(state)
        else:
            self.sample_ids, self.state = ids.to(device), state.to(device)
        return self

    def unkey(self):
        self.sample_ids = self.state = None
        return self

    @property
    def keyed(self):
        return self.sample_ids is not None

    def graph_key(self):
        """what a captured launch of this generator bakes in"""
        return None if not self.keyed else (self.sample_ids.data_ptr(), self.state.data_ptr(), int(self.sample_ids.shape[0]))

    def snapshot(self, device):
        """state to put back with ``restore`` (the eager warm-up in front of a graph capture must not consume draws)"""
        if self.keyed:
            return ('indexed', self.state.clone())
        return ('stream', torch.cuda.get_rng_state(device))

    def restore(self, snap, device):
        if snap[0] == 'indexed':
            self.state.copy_(snap[1])
        else:
            torch.cuda.set_rng_state(snap[1], device)

    def ids_for(self, n):
        """the sample ids of a draw with n rows: all of them, or — a short last batch of a loader that keeps it, as the
        reference's DataLoader does (DDP:1746-1752) — the first n (a view: same storage, so a captured graph keyed on the
        full-size batch is not disturbed)"""
        n = int(n)
        if n > self.sample_ids.shape[0]:
            raise ValueError(f'the generator is keyed for {self.sample_ids.shape[0]} rows, a draw asks for {n}: '
                             f'key_by_sample with the ids of this batch first')
        return self.sample_ids if n == self.sample_ids.shape[0] else self.sample_ids[:n]

    def randn(self, shape, device):
        if self.keyed:
            return ops.rng_indexed(shape, self.ids_for(shape[0]), self.state, 0)
        return torch.randn(tuple(shape), device=device)

    def uniform(self, n, device):
        if self.keyed:
            return ops.rng_indexed((n,), self.ids_for(n), self.state, 1)
        return torch.zeros((n,), device=device).float().uniform_(0, 1)


# development knob (same-box A/Bs): '0' = the replayed step computes its embeddings and (scale, shift) rows per step
SS_TABLES = __import__('os').environ.get('DMH_SS_TABLES', '1') != '0'


class Unet(nn.Module):
    """CFG:302-466.  ``forward`` launches the HIP program; parameters live in holders."""

    def __init__(self, dim, num_classes, cond_drop_prob=0.5, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8),
                 channels=3, resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16):
        super().__init__()
        self.cond_drop_prob = cond_drop_prob
        self.channels = channels
        input_channels = channels + 3                       # rgb_flow * mask is concatenated, CFG:330,430
        init_dim = default(init_dim, dim)
        self.init_conv = nn.Conv2d(input_channels, init_dim, 7, padding=3)
        time_dim = dim * 4
        self.random_or_learned_sinusoidal_cond = learned_sinusoidal_cond or random_fourier_features
        # (GaussianDiffusion refuses such a model — CFG:514-515 — so it serves bare Unet.forward callers only)
        pos_emb, fourier_dim = P.time_embedding(dim, learned_sinusoidal_cond, random_fourier_features, learned_sinusoidal_dim)
        self.time_mlp = nn.Sequential(pos_emb, nn.Linear(fourier_dim, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        self.classes_emb = nn.Embedding(num_classes, dim)
        self.null_classes_emb = nn.Parameter(torch.randn(dim))
        classes_dim = dim * 4
        self.classes_mlp = nn.Sequential(nn.Linear(dim, classes_dim), nn.GELU(), nn.Linear(classes_dim, classes_dim))
        self.out_dim = default(out_dim, channels * (1 if not learned_variance else 2))
        P.build_trunk(self, dim, init_dim, dim_mults, input_channels, time_dim + classes_dim, resnet_block_groups,
                      self.out_dim, P.downsample_cfg)
        self.rng = DeviceRng()
        self._engine = UnetEngine(self, groups=resnet_block_groups)

    # ---- class-dropout draw of CFG:421-425 / prob_mask_like CFG:84-90
    def _keep_mask(self, batch, cond_drop_prob, device):
        if not cond_drop_prob > 0:
            return None
        prob = 1 - cond_drop_prob
        if prob == 1:
            return None                                      # keep every row
        if prob == 0:
            return self._null_mask(batch, device)
        if type(self.rng) is DeviceRng and self.rng.keyed:           # draw + compare + uint8 in one launch
            return ops.rng_keep_mask(self.rng.ids_for(batch), self.rng.state, prob)
        return (self.rng.uniform(batch, device) < prob).to(torch.uint8)

    def _null_mask(self, batch, device):
        """(batch,) uint8 zeros: every class dropped (the null pass, CFG:409) — one buffer per (batch, device), allocated once,
        never written and never replaced: a captured denoise step holds its address for as long as the graph lives"""
        cache = self.__dict__.setdefault('_null_keep', {})
        key = (int(batch), str(torch.device(device)))
        z = cache.get(key)
        if z is None:
            z = cache[key] = torch.zeros((batch,), device=device, dtype=torch.uint8)
        return z

    def _stem(self, x, rgb_flow, mask):
        """cat((x, rgb_flow*mask)) -> NHWC -> init_conv: identical for the cond and the null pass (CFG:430-432)."""
        if not x.is_cuda:
            raise RuntimeError('dmhomo_amd.Unet runs on the GPU only (HIP kernels); move the inputs with .cuda()')
        eng = self._engine
        eng.ensure_prepared()
        xin = ops.assemble_input(x.to(torch.float32).contiguous(), rgb_flow.to(torch.float32).contiguous(),
                                 mask.to(torch.float32).contiguous(), reps=1, cpad=eng.cin_pad)
        return eng.stem(xin)

    def _run(self, x, time, classes, rgb_flow, mask, keeps, taps=None, x0=None, first=None, out=None, rows=None):
        """rows [rep*B + b]: sample b under class-keep mask keeps[rep] -> (len(keeps)*B, out_dim, H, W).
        first: engine.first_conv(x0) when the caller shares it between passes.
        rows: ``ops.rows_from_keep`` list of the rows to compute (the others stay unwritten), or None for all."""
        eng = self._engine
        if x0 is None:
            x0 = self._stem(x, rgb_flow, mask)
        if len(keeps) > 1:
            x0 = x0.repeat(len(keeps), 1, 1, 1)              # row copies of the shared stem output
        time = time.to(torch.int64).contiguous()
        classes = classes.to(torch.int64).contiguous()
        tab = self.__dict__.get('_ss_tab')
        if tab is not None:
            # a replayed denoise step (GaussianDiffusion._sample_graphed): the (scale, shift) rows of every ResnetBlock come
            # from tables indexed by the step cursor and the row's class / keep bit — one small launch per pass instead of the
            # two embeddings, four small linears and the 512 -> 8 k linear at its head; bitwise the same rows (UnetEngine.ss_tables)
            T, Ct, cursor = tab
            B = classes.shape[0]
            ss_all = torch.empty((len(keeps) * B, T.shape[1]), device=T.device, dtype=torch.float32)
            for r, k in enumerate(keeps):
                ops.ss_gather(T, Ct, eng.mlp_b, cursor, classes, k, out=ss_all[r * B:(r + 1) * B])
            return eng.trunk(x0, None, taps, first=first, out=out, ss_all=ss_all, rows=rows)
        cond = eng.embed(time, [(classes, k) for k in keeps], len(keeps))
        return eng.trunk(x0, cond, taps, first=first, out=out, rows=rows)

    def forward(self, x, time, classes, rgb_flow, mask, cond_drop_prob=None):
        cond_drop_prob = default(cond_drop_prob, self.cond_drop_prob)
        keep = self._keep_mask(x.shape[0], cond_drop_prob, x.device)
        return self._run(x, time, classes, rgb_flow, mask, [keep])

    # 'batched': cond + null rows as ONE 2B launch sequence.  'streams': the two passes as two B-row
    # sequences on two HIP streams, so one pass's HBM-bound kernels and kernel tails overlap the other's
    # matrix-bound kernels.  Results are identical (rows are independent, tests pin that bitwise).
    cfg_mode = 'batched'
    stream_splits = 1      # 'streams' mode: row sub-batches per pass, each on its own stream
    # 'streams' mode: downs.0.0.block1.proj(stem output) once for both passes (bitwise the same rows); DMH_SHARE_FIRST_CONV=0
    # is the development knob for same-box A/Bs
    share_first_conv = __import__('os').environ.get('DMH_SHARE_FIRST_CONV', '1') != '0'

    def _cond_null(self, x, time, classes, rgb_flow, mask):
        """the two passes of CFG:404,409: (cond logits, null logits, computed).  computed is None — every row of the cond
        logits was computed — or, with ``dedup_dropped_rows``, the (B,) uint8 class-keep mask: rows where it is 0 were left
        UNWRITTEN in the cond logits and equal the null logits' rows (``ops.sampler_step(..., keep=computed)``)."""
        B = x.shape[0]
        keep = self._keep_mask(B, self.cond_drop_prob, x.device)
        null = self._null_mask(B, x.device)
        dedup = bool(self.dedup_dropped_rows) and keep is not None
        if self.cfg_mode == 'streams':
            nsub = max(1, min(int(self.stream_splits), B))       # row sub-batches per pass (each on its own stream)
            nstreams = 2 * nsub
            if len(getattr(self, '_side', ())) != nstreams:
                self._side = tuple(torch.cuda.Stream(device=x.device) for _ in range(nstreams))
            cur = torch.cuda.current_stream()
            x0 = self._stem(x, rgb_flow, mask)                # once, on the main stream
            # ... and with it the first convolution behind it: the class embedding reaches a ResnetBlock only through the
            # scale / shift behind block1's GroupNorm, so downs.0.0.block1.proj(x0) is the same rows in both passes
            first = self._engine.first_conv(x0) if self.share_first_conv else None
            cond_out = torch.empty((B, self.out_dim) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
            null_out = torch.empty_like(cond_out)
            bounds = [(i * B) // nsub for i in range(nsub + 1)]
            si = 0
            for k, dst, sub in ((keep, cond_out, dedup), (null, null_out, False)):
                for lo, hi in zip(bounds[:-1], bounds[1:]):
                    st = self._side[si]
                    si += 1
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        kk = None if k is None else k[lo:hi].contiguous()
                        fs = None if first is None else (first[0][lo:hi], first[1][lo:hi])
                        # (the pass writes its rows of the result itself: the fused final projection's destination)
                        self._run(None, time[lo:hi], classes[lo:hi], None, None, [kk], x0=x0[lo:hi], first=fs, out=dst[lo:hi],
                                  rows=ops.rows_from_keep(kk) if sub else None)
            for st in self._side:
                cur.wait_stream(st)
            return cond_out, null_out, (keep if dedup else None)
        both = self._run(x, time, classes, rgb_flow, mask, [keep, null],
                         rows=ops.rows_from_keep(keep, extra=B) if dedup else None)
        return both[:B], both[B:], (keep if dedup else None)

    # OPT-IN, off by default (bench.py's headline keeps it off and reports it under "variants").  The reference's conditional
    # pass draws a class-dropout mask with p = 0.5 (CFG:404 -> CFG:415,422), also while sampling: a dropped row of that pass
    # has exactly the inputs of the same sample's row in the null pass, so its logits equal the null logits and the guided
    # output is the null output.  With this switch those duplicate rows are not computed: the mask never leaves the device —
    # ``ops.rows_from_keep`` turns it into the list of active rows that every launch of the conditional pass takes
    # (include/dmhomo_hip.h, "Row subsets": the captured B-row grid stays, workgroups of inactive rows retire at once), and the
    # sampler step reads the null logits where the mask is 0.  B + (kept rows) UNet rows per step instead of 2B, the result
    # bit for bit the same (rows are independent of their batch — tests pin that), in every cfg_mode and under hip_graph.
    dedup_dropped_rows = False

    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):
        """CFG:403-410: ``forward(*args, **kwargs)``; for cond_scale != 1 blended with ``forward(*args, cond_drop_prob=1.,
        **kwargs)`` — so, as in the reference, a caller's own ``cond_drop_prob`` is accepted only with cond_scale == 1."""
        if cond_scale == 1:
            return self.forward(*args, **kwargs)
        if 'cond_drop_prob' in kwargs or len(args) > 5:
            raise TypeError("forward() got multiple values for keyword argument 'cond_drop_prob'")   # as CFG:409 would
        x, time, classes, rgb_flow, mask = _bind_forward(self.forward, args, kwargs)
        logits, null, computed = self._cond_null(x, time, classes, rgb_flow, mask)
        step = DmhStep(objective=ops.OBJECTIVE['pred_x0'], clip=0, mode=ops.MODE_LAST, cond_scale=float(cond_scale),
                       sqrt_recip_ac=1., sqrt_recipm1_ac=1.)
        out, _, _ = ops.sampler_step(step, logits, null, null, None, want_x_start=False, keep=computed)
        return out


def _bind_forward(forward, args, kwargs):
    import inspect
    ba = inspect.signature(forward).bind(*args, **kwargs)           # TypeError for missing / unknown arguments, like a call
    a = ba.arguments
    return a['x'], a['time'], a['classes'], a['rgb_flow'], a['mask']


class ScheduleHost:
    """host mirrors of the schedule buffers: the reference indexes device buffers with python ints and does
    0-dim fp32 tensor arithmetic on them (CFG:697-701); here the same ops run on CPU copies and the results
    enter the sampler kernel as scalars."""

    _HOST_NAMES = ('alphas_cumprod', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'sqrt_alphas_cumprod',
                   'sqrt_one_minus_alphas_cumprod', 'posterior_mean_coef1', 'posterior_mean_coef2',
                   'posterior_log_variance_clipped')

    def _host(self):
        """CPU copies of the schedule buffers, cached per buffer version (a device -> host copy per sampling call would
        also be illegal inside a HIP-graph capture)."""
        sig = tuple((getattr(self, n).data_ptr(), getattr(self, n)._version) for n in self._HOST_NAMES)
        cache = self.__dict__.get('_host_cache')
        if cache is None or cache[0] != sig:
            cache = (sig, {n: getattr(self, n).detach().cpu() for n in self._HOST_NAMES})
            self.__dict__['_host_cache'] = cache
        return cache[1]

    # ---- D4 / D7: the affine combinations of CFG:586-608 [DDP:584-611] with a timestep per row
    def _at(self, name, t, neg=False):
        a = getattr(self, name)
        return (-a if neg else a).gather(-1, t.to(torch.int64)).contiguous()

    def predict_start_from_noise(self, x_t, t, noise):
        """CFG:586-588: extract(sqrt_recip_ac) * x_t - extract(sqrt_recipm1_ac) * noise."""
        return ops.rows_lincomb(x_t, self._at('sqrt_recip_alphas_cumprod', t), noise,
                                self._at('sqrt_recipm1_alphas_cumprod', t, neg=True))

    def predict_noise_from_start(self, x_t, t, x0):
        """CFG:590-594: (extract(sqrt_recip_ac) * x_t - x0) / extract(sqrt_recipm1_ac)."""
        minus_one = torch.full((x_t.shape[0],), -1., device=x_t.device, dtype=torch.float32)
        return ops.rows_lincomb(x_t, self._at('sqrt_recip_alphas_cumprod', t), x0, minus_one,
                                div=self._at('sqrt_recipm1_alphas_cumprod', t))

    def predict_v(self, x_start, t, noise):
        """CFG:596-598: extract(sqrt_ac) * noise - extract(sqrt_1m_ac) * x_start."""
        return ops.rows_lincomb(noise, self._at('sqrt_alphas_cumprod', t), x_start,
                                self._at('sqrt_one_minus_alphas_cumprod', t, neg=True))

    def predict_start_from_v(self, x_t, t, v):
        """CFG:600-601: extract(sqrt_ac) * x_t - extract(sqrt_1m_ac) * v."""
        return ops.rows_lincomb(x_t, self._at('sqrt_alphas_cumprod', t), v,
                                self._at('sqrt_one_minus_alphas_cumprod', t, neg=True))

    def q_posterior(self, x_start, x_t, t):
        """CFG:603-608: (posterior mean, variance, clipped log variance), the last two shaped (b, 1, 1, 1)."""
        mean = ops.rows_lincomb(x_start, self._at('posterior_mean_coef1', t), x_t, self._at('posterior_mean_coef2', t))
        return (mean, extract(self.posterior_variance, t, x_t.shape),
                extract(self.posterior_log_variance_clipped, t, x_t.shape))

    def _predictions_per_row(self, model_output, x, t, clip_x_start):
        """the objective branch of model_predictions (CFG:614-628) for a batch whose rows sit at different timesteps"""
        one = torch.ones((x.shape[0],), device=x.device, dtype=torch.float32)
        clip = (lambda v: ops.rows_lincomb(v, one, clamp=True)) if clip_x_start else (lambda v: v)
        if self.objective == 'pred_noise':
            pred_noise = model_output
            x_start = clip(self.predict_start_from_noise(x, t, pred_noise))
        elif self.objective == 'pred_x0':
            x_start = clip(model_output)
            pred_noise = self.predict_noise_from_start(x, t, x_start)
        else:                                                # pred_v
            x_start = clip(self.predict_start_from_v(x, t, model_output))
            pred_noise = self.predict_noise_from_start(x, t, x_start)
        return ModelPrediction(pred_noise, x_start)

    def _ddim_coef(self, host, time, time_next):
        """sqrt(alpha_next), c, sigma in the reference's op order, CFG:697-701."""
        alpha = host['alphas_cumprod'][time]
        alpha_next = host['alphas_cumprod'][time_next]
        sigma = self.ddim_sampling_eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        # For a first jump from t=T-1 (alpha ~ 2e-9) to alpha_next >~ 1e-2 (s_step <= 8) the radicand is pure fp32
        # cancellation noise of +-6e-8 and the reference yields c = NaN on hosts whose sqrt rounds the other way
        # (seen: 999 -> 499).  Clamping at 0 changes nothing where the reference is finite.
        c = (1 - alpha_next - sigma ** 2).clamp(min=0).sqrt()
        return float(alpha_next.sqrt()), float(c), float(sigma)


class GaussianDiffusion(nn.Module, ScheduleHost):
    """CFG:498-842 — sampling side on the GPU kernels; buffers and their names as the reference."""

    def __init__(self, model, *, image_size, timesteps=1000, sampling_timesteps=None, loss_type='l1',
                 objective='pred_noise', beta_schedule='cosine', p2_loss_weight_gamma=0., p2_loss_weight_k=1,
                 ddim_sampling_eta=1.):
        super().__init__()
        assert not (type(self) == GaussianDiffusion and model.channels != model.out_dim)
        assert not model.random_or_learned_sinusoidal_cond
        self.model = model
        self.channels = self.model.channels
        self.image_size = image_size
        self.objective = objective
        assert objective in {'pred_noise', 'pred_x0', 'pred_v'}, \
            'objective must be either pred_noise (predict noise) or pred_x0 (predict image start) or pred_v (predict v)'
        bufs = make_buffers(beta_schedule, timesteps, p2_loss_weight_gamma, p2_loss_weight_k)
        self.num_timesteps = int(bufs['betas'].shape[0])
        self.loss_type = loss_type
        self.sampling_timesteps = default(sampling_timesteps, timesteps)
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        for name, val in bufs.items():
            self.register_buffer(name, val)

    @property
    def rng(self):
        return self.model.rng

    @rng.setter
    def rng(self, value):
        self.model.rng = value

    def _step(self, host, t, mode, cond_scale, clip, c=(0., 0., 0.)):
        return DmhStep(objective=ops.OBJECTIVE[self.objective], clip=int(bool(clip)), mode=mode,
                       cond_scale=float(cond_scale),
                       sqrt_recip_ac=float(host['sqrt_recip_alphas_cumprod'][t]),
                       sqrt_recipm1_ac=float(host['sqrt_recipm1_alphas_cumprod'][t]),
                       sqrt_ac=float(host['sqrt_alphas_cumprod'][t]),
                       sqrt_1m_ac=float(host['sqrt_one_minus_alphas_cumprod'][t]),
                       c0=float(c[0]), c1=float(c[1]), c2=float(c[2]))

    @staticmethod
    def _uniform_time(t):
        """python int when every row shares the timestep (always true while sampling), else None"""
        t0 = int(t[0])
        if t.numel() > 1 and not bool((t == t0).all()):
            return None
        return t0

    def _network(self, x, t, classes, rgb_flow, mask, cond_scale):
        """-> (cond logits, null logits or None, computed-rows mask or None): see Unet._cond_null"""
        if cond_scale == 1:
            return self.model.forward(x, t, classes, rgb_flow, mask), None, None
        return self.model._cond_null(x, t, classes, rgb_flow, mask)

    def model_predictions(self, x, t, classes, rgb_flow, mask, cond_scale=3., clip_x_start=False):
        """CFG:610-630.  One timestep for the whole batch (what the samplers pass): blend, objective branch and clamp in
        ONE pass of dmh_sampler_step; a timestep per row (p_losses-style callers): the same arithmetic row by row."""
        host = self._host()
        cond, null, computed = self._network(x, t, classes, rgb_flow, mask, cond_scale)
        t0 = self._uniform_time(t)
        if t0 is None:
            if null is not None:                             # null + (cond - null) * cond_scale, CFG:410
                blend = DmhStep(objective=ops.OBJECTIVE['pred_x0'], clip=0, mode=ops.MODE_LAST,
                                cond_scale=float(cond_scale), sqrt_recip_ac=1., sqrt_recipm1_ac=1.)
                cond, _, _ = ops.sampler_step(blend, cond, null, null, None, want_x_start=False, keep=computed)
            return self._predictions_per_row(cond, x.contiguous(), t, clip_x_start)
        step = self._step(host, t0, ops.MODE_LAST, cond_scale, clip_x_start)
        _, x_start, pred_noise = ops.sampler_step(step, cond, null, x.contiguous(), None, True, True, keep=computed)
        return ModelPrediction(pred_noise, x_start)

    def p_mean_variance(self, x, t, classes, cond_scale, clip_denoised=True):
        """CFG:632-637 as it stands: it hands model_predictions ``(x, t, classes, cond_scale)`` — cond_scale in rgb_flow's
        place and no mask — so the call raises TypeError.  Kept failing at the same call."""
        preds = self.model_predictions(x, t, classes, cond_scale)
        x_start = preds.pred_x_start
        if clip_denoised:
            x_start.clamp_(-1., 1.)
        model_mean, posterior_variance, posterior_log_variance = self.q_posterior(x_start=x_start, x_t=x, t=t)
        return model_mean, posterior_variance, posterior_log_variance, x_start

    @torch.no_grad()
    def p_sample(self, x, t: int, classes, cond_scale=3., clip_denoised=True):
        """CFG:639-654: one ancestral step through p_mean_variance — which raises (above), as in the reference."""
        batched_times = torch.full((x.shape[0],), t, device=x.device, dtype=torch.long)
        model_mean, _, model_log_variance, x_start = self.p_mean_variance(x=x, t=batched_times, classes=classes,
                                                                          cond_scale=cond_scale,
                                                                          clip_denoised=clip_denoised)
        b = x.shape[0]
        one = torch.ones((b,), device=x.device, dtype=torch.float32)
        if not t > 0:
            return model_mean, x_start
        noise = self.rng.randn(x.shape, x.device)
        return ops.rows_lincomb(model_mean, one, noise, (0.5 * model_log_variance).exp().reshape(b)), x_start

    @torch.no_grad()
    def ddim_sample(self, classes, rgb_flow, flow, mask, shape, cond_scale=3., clip_denoised=True):
        """CFG:669-711."""
        return self._ddim_sample(classes, rgb_flow, flow, mask, shape, cond_scale, clip_denoised)

    def _ddim_sample(self, classes, rgb_flow, flow, mask, shape, cond_scale=3., clip_denoised=True, trace=None):
        """ddim_sample; ``trace`` (list) optionally receives per-step x_start / img for parity tests."""
        batch, device = shape[0], self.betas.device
        host = self._host()
        img = self.rng.randn(shape, device).contiguous()
        for time, time_next in ddim_pairs(self.num_timesteps, self.sampling_timesteps):
            time_cond = torch.full((batch,), time, device=device, dtype=torch.long)
            cond, null, computed = self._network(img, time_cond, classes, rgb_flow, mask, cond_scale)
            if time_next < 0:
                step = self._step(host, time, ops.MODE_LAST, cond_scale, clip_denoised)
                noise = None
            else:
                step = self._step(host, time, ops.MODE_DDIM, cond_scale, clip_denoised,
                                  self._ddim_coef(host, time, time_next))
                noise = self.rng.randn(shape, device).contiguous()
            img, x_start, _ = ops.sampler_step(step, cond, null, img, noise, want_x_start=trace is not None, keep=computed)
            if trace is not None:
                trace.append({'time': time, 'x_start': x_start, 'img': img})
        img = ops.affine(img, 0.5, 0.5)                      # unnormalize_to_zero_to_one, CFG:709
        return img, mask, flow

    @torch.no_grad()
    def p_sample_loop(self, classes, shape, cond_scale=3.):
        # CFG:656 takes (classes, shape, cond_scale) while CFG:719-720 calls it with six arguments, and
        # CFG:633 mis-passes p_mean_variance's arguments: the reference's ancestral path cannot run
        # (SURVEY.md fact 6).  Kept failing in the same way rather than inventing semantics.
        raise TypeError('classifier_free_guidance.GaussianDiffusion.p_sample_loop is not callable in the reference '
                        '(CFG:656 vs CFG:719-720); use sampling_timesteps < timesteps (DDIM), or the unconditional '
                        'denoising_diffusion_pytorch.GaussianDiffusion for ancestral sampling')

    # hip_graph = True: ONE denoise step of the sampling loop of CFG:683-707 (every kernel of it, on however many HIP
    # streams cfg_mode uses) is captured per (shapes, cond_scale, weight version, schedule) into a HIP graph and replayed
    # S times: the step's coefficients and its timestep come from device tables that a cursor kernel at the end of the
    # step advances (dmh_sampler_seek), so the same ~660-node graph serves every step and any S (s_step = 250 included),
    # the capture costs one step, and the host queues replay k + 1 while replay k runs.  The last step (no noise draw,
    # img = x_start, CFG:693-695) is a second graph in the same memory pool.  Same kernels, same order, same device RNG
    # stream (the Philox offsets are graph inputs): results are bitwise those of the eager path, the capturing call
    # included (the RNG state is restored after the eager warm-up).  Off by default; bench.py switches it on.
    # Captured steps are kept in a small least-recently-used cache (``graph_cache_size`` entries, each with its own graphs,
    # memory pool and static buffers): a job that alternates batch shapes — the short last batch of every epoch of
    # scripts/dgm_sample.py's loader, two samplers sharing one model — captures each shape once, not once per alternation.
    hip_graph = False
    graph_cache_size = 4
    graph_captures = 0                   # captures made by this object so far (tests count them)

    @torch.no_grad()
    def sample(self, classes, rgb_flow, flow, mask, cond_scale=3.):
        """CFG:713-720."""
        batch_size, image_size, channels = classes.shape[0], self.image_size, self.channels
        shape = (batch_size, channels, image_size, image_size)
        if not self.is_ddim_sampling:
            return self.p_sample_loop(classes, rgb_flow, flow, mask, shape, cond_scale)    # TypeError, as CFG:719-720
        if self.hip_graph and type(self.rng) is DeviceRng and self.sampling_timesteps >= 1:
            return self._sample_graphed(classes, rgb_flow, flow, mask, shape, cond_scale)
        rgb_flow = ops.affine(rgb_flow.to(torch.float32), 2., -1.)      # normalize_to_neg_one_to_one, CFG:716
        return self.ddim_sample(classes, rgb_flow, flow, mask, shape, cond_scale)

    def _sample_graphed(self, classes, rgb_flow, flow, mask, shape, cond_scale):
        eng = self.model._engine
        eng.ensure_prepared()
        host = self._host()                                  # (host mirrors cached before any capture)
        device = classes.device
        clip = True                                          # ddim_sample's clip_denoised default, as sample() calls it
        # everything that is baked into the captured launches or into the step tables
        key = (tuple(shape), tuple(rgb_flow.shape), float(cond_scale), self.model.cfg_mode, int(self.model.stream_splits),
               bool(self.model.dedup_dropped_rows), bool(self.model.share_first_conv), float(self.model.cond_drop_prob), eng._sig, self.sampling_timesteps,
               self.num_timesteps, self.objective, float(self.ddim_sampling_eta), clip, self.__dict__['_host_cache'][0],
               str(device), self.rng.graph_key())
        cache = self.__dict__.setdefault('_graph_states', OrderedDict())
        st = cache.get(key)
        if st is not None:
            cache.move_to_end(key)
        else:
            steps, times = [], []
            for time, time_next in ddim_pairs(self.num_timesteps, self.sampling_timesteps):
                if time_next < 0:
                    steps.append(self._step(host, time, ops.MODE_LAST, cond_scale, clip))
                else:
                    steps.append(self._step(host, time, ops.MODE_DDIM, cond_scale, clip,
                                            self._ddim_coef(host, time, time_next)))
                times.append(time)
            assert steps[-1].mode == ops.MODE_LAST and all(s_.mode == ops.MODE_DDIM for s_ in steps[:-1])
            table, tt, cursor, cur = ops.step_table(steps, times, device)
            ins = [classes.clone(), rgb_flow.to(torch.float32).clone(), mask.clone()]
            st = {'key': key, 'ins': ins, 'table': table, 'times': tt, 'cursor': cursor, 'cur': cur, 'nsteps': len(steps),
                  'rf': torch.empty_like(ins[1]), 'img': torch.zeros(shape, device=device),
                  'tcond': torch.zeros((shape[0],), device=device, dtype=torch.long)}

            # the embedding side of the network depends on (step, class, keep bit) only: tables, made once per capture
            if SS_TABLES:
                T_tab, C_tab = eng.ss_tables(times)
                st['ss_tab'] = (T_tab, C_tab, cursor)

            def mid():                                       # one denoise step of CFG:684-707, in place on st['img']
                cond, null, computed = self._network(st['img'], st['tcond'], ins[0], st['rf'], ins[2], cond_scale)
                noise = self.rng.randn(shape, device).contiguous()
                ops.sampler_step_dev(cur, cond, null, st['img'], noise, out=st['img'], keep=computed)
                ops.sampler_seek(cursor, -1, table, tt, cur, st['tcond'])

            def last():                                      # CFG:693-695 + unnormalize, CFG:709
                cond, null, computed = self._network(st['img'], st['tcond'], ins[0], st['rf'], ins[2], cond_scale)
                x0 = ops.sampler_step_dev(cur, cond, null, st['img'], None, keep=computed)
                return ops.affine(x0, 0.5, 0.5)
            # eager warm-up of both bodies on a side stream (first-launch work: LDS attributes, side streams), as
            # torch.cuda.graphs asks for; the device RNG state is put back afterwards, so the capturing call consumes
            # exactly what an eager sample() would
            rng_state = self.rng.snapshot(device)
            ops.affine(ins[1], 2., -1., out=st['rf'])
            ops.sampler_seek(cursor, 0, table, tt, cur, st['tcond'])
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream())
            self.model.__dict__['_ss_tab'] = st.get('ss_tab')         # (read by Unet._run while the step bodies run)
            try:
                with torch.cuda.stream(side):
                    mid()
                    ops.sampler_seek(cursor, len(steps) - 1, table, tt, cur, st['tcond'])   # last() runs on the last entry
                    last()
                torch.cuda.current_stream().wait_stream(side)
                g_mid = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_mid, capture_error_mode='thread_local'):
                    mid()
                g_last = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_last, pool=g_mid.pool(), capture_error_mode='thread_local'):
                    st['out'] = last()
            finally:
                self.model.__dict__.pop('_ss_tab', None)
            self.rng.restore(rng_state, device)
            st['graph'], st['graph_last'] = g_mid, g_last
            cache[key] = st
            self.graph_captures = self.graph_captures + 1
            while len(cache) > max(int(self.graph_cache_size), 1):
                cache.popitem(last=False)                    # least recently used: its graphs, pool and buffers go with it
        self.__dict__['_graph_state'] = st                   # (the entry this call replays)
        for dst, src in zip(st['ins'], (classes, rgb_flow, mask)):
            dst.copy_(src)
        ops.affine(st['ins'][1], 2., -1., out=st['rf'])      # normalize_to_neg_one_to_one, CFG:716
        st['img'].copy_(self.rng.randn(shape, device))       # CFG:679
        ops.sampler_seek(st['cursor'], 0, st['table'], st['times'], st['cur'], st['tcond'])
        for _ in range(st['nsteps'] - 1):
            st['graph'].replay()
        st['graph_last'].replay()
        return st['out'].clone(), mask, flow

    @torch.no_grad()
    def interpolate(self, x1, x2, t=None, lam=0.5):
        """CFG:722-736 calls ``self.p_sample(img, t)`` — two arguments for a method that needs classes and conditions
        (CFG:639-654), and p_sample itself mis-calls p_mean_variance: it raises TypeError in the
        reference for any t > 0, and so does this."""
        raise TypeError('classifier_free_guidance.GaussianDiffusion.interpolate is not callable in the reference '
                        '(CFG:733 vs CFG:639); use denoising_diffusion_pytorch.GaussianDiffusion.interpolate')

    def q_sample(self, x_start, t, noise=None):
        """CFG:738-742."""
        noise = default(noise, lambda: self.rng.randn(x_start.shape, x_start.device))
        ca = self.sqrt_alphas_cumprod.gather(-1, t).contiguous()
        cb = self.sqrt_one_minus_alphas_cumprod.gather(-1, t).contiguous()
        return ops.q_sample(x_start.contiguous(), noise.contiguous(), ca, cb)

    @property
    def loss_fn(self):
        """CFG:744-751 — the name of the elementwise loss (the reduction runs in dmh_diff_mean)."""
        if self.loss_type in ('l1', 'l2'):
            return self.loss_type
        raise ValueError(f'invalid loss type {self.loss_type}')

    @torch.no_grad()
    def p_losses(self, x_start, t, *, classes, rgb_flow, flow, mask, noise=None):
        """CFG:770-806, the loss VALUE: q_sample -> UNet (class dropout p=0.5) -> flow_warp -> L1/L2 + alpha_bar_t-weighted
        masked photometric term.  No autograd graph is built here; ``forward`` returns a loss with a grad_fn whose
        gradients come from the HIP backward kernels of dmhomo_amd.train (same forward, activations saved)."""
        from .ddpm import flow_warp
        squared = self.loss_fn == 'l2'
        noise = default(noise, lambda: self.rng.randn(x_start.shape, x_start.device))
        x_start = x_start.to(torch.float32).contiguous()
        noise = noise.to(torch.float32).contiguous()
        t = t.to(torch.int64).contiguous()
        x = self.q_sample(x_start, t, noise)
        model_out = self.model(x, t, classes, rgb_flow=rgb_flow, mask=mask)
        im1, im2 = model_out[:, :3].contiguous(), model_out[:, 3:].contiguous()
        im2_warp = flow_warp(im2, flow)
        if self.objective == 'pred_noise':
            target = noise
        elif self.objective == 'pred_x0':
            target = x_start
        elif self.objective == 'pred_v':                         # predict_v, CFG:596-598
            ca = self.sqrt_alphas_cumprod.gather(-1, t).contiguous()
            cb = (-self.sqrt_one_minus_alphas_cumprod).gather(-1, t).contiguous()
            target = ops.q_sample(noise, x_start, ca, cb)
        else:
            raise ValueError(f'unknown objective {self.objective}')
        loss = ops.diff_mean(model_out, target, None, squared)
        photo = ops.diff_mean(im2_warp, im1, mask.to(torch.float32).contiguous(), squared)
        w = self.alphas_cumprod.gather(-1, t).contiguous()
        return ops.loss_combine(loss, photo, w)

    def forward(self, img, *args, **kwargs):
        """CFG:808-842: split the 12-channel batch (DDP:1162 layout), draw t, evaluate p_losses.

        With autograd enabled and trainable parameters the returned loss carries a grad_fn: ``loss.backward()`` fills
        ``parameter.grad`` exactly as the reference's does, so a user-written loop (``loss = diffusion(batch,
        classes=c); loss.backward(); opt.step()``, DDP:1843-1857) runs unchanged — the gradients come from the HIP
        backward kernels of dmhomo_amd.train (no autograd graph inside), computed together with the loss.  Under
        ``torch.no_grad()`` only the value is evaluated."""
        classes = kwargs.get('classes', args[0] if args else None)
        if torch.is_grad_enabled() and classes is not None and any(p.requires_grad for p in self.model.parameters()):
            from .train import loss_with_grad_fn
            return loss_with_grad_fn(self, img, classes)
        b, c, h, w = img.shape
        assert h == self.image_size and w == self.image_size, f'height and width of image must be {self.image_size}'
        t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        data = ops.affine(img[:, :6].to(torch.float32), 2., -1.)
        mask = img[:, 6:7].contiguous()
        rgb_flow = ops.affine(img[:, -5:-2].to(torch.float32), 2., -1.)
        flow = img[:, -2:].contiguous()
        return self.p_losses(data, t, *args, rgb_flow=rgb_flow, flow=flow, mask=mask, **kwargs)

"""Multi-GPU sampling: one process per GPU, samples sharded across ranks.

The denoising path has no cross-sample operation (GroupNorm, attention and the DDIM update
are all per sample), so a global batch splits contiguously over the ranks with NO data-path
collective.  RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in CPU tests) is used for
  (1) one broadcast of the UNet weights from rank 0 at start-up — issued as scatter +
      all-gather so that on the point-to-point xGMI mesh every link carries 1/world of the
      153.7 MB payload instead of one link carrying all of it;
  (2) one gather per batch of the uint8 image pairs + 3x3 homographies to rank 0.
The reference's counterpart is N hand-launched independent processes (README:14,
dgm_sample.py:13-18) that each load the checkpoint from disk.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # an explicit backend='gloo' ARGUMENT asks for a host-only process group (the CPU plumbing tests), also on a box with
    # GPUs; DMH_DIST_BACKEND=gloo (environment) keeps the ranks on the GPU and only swaps the transport
    use_cuda = torch.cuda.is_available() and backend != 'gloo'
    if use_cuda:
        # DMH_SHARE_GPU=1 (tests on a one-GPU box): the ranks of the job time-share the visible GPUs; RCCL refuses two
        # ranks on one device, so such a job runs over gloo (DMH_DIST_BACKEND=gloo; the collectives then stage through host memory)
        if os.environ.get('DMH_SHARE_GPU') == '1':
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
    backend = backend or os.environ.get('DMH_DIST_BACKEND') or ('nccl' if use_cuda else 'gloo')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend, rank=rank, world_size=world)
    device = torch.device('cuda', local) if use_cuda else torch.device('cpu')
    return rank, world, device


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def _host_staged():
    """gloo moves device tensors only for broadcast / all_reduce: its scatter / gather / all_gather take host tensors"""
    return dist.is_initialized() and dist.get_backend() == 'gloo'


def _to_coll(t):
    return t.cpu() if (_host_staged() and t.is_cuda) else t


def shard_bounds(total, rank, world):
    """contiguous [lo, hi) of ``total`` samples owned by ``rank`` (first ``total % world`` ranks get one more)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@torch.no_grad()
def broadcast_module_(module, src=0):
    """make every rank's parameters and buffers equal to ``src``'s with ONE payload:
    flatten -> scatter (1/world per peer) -> all-gather -> unflatten."""
    if world_size() == 1:
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
    others = [t for t in module.buffers() if not t.is_floating_point()]
    flat = _to_coll(torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors]))
    n = flat.numel()
    per = (n + world - 1) // world
    padded = torch.zeros(per * world, device=flat.device, dtype=torch.float32)
    padded[:n] = flat
    mine = torch.empty(per, device=flat.device, dtype=torch.float32)
    chunks = list(padded.split(per)) if rank == src else None
    dist.scatter(mine, chunks, src=src)
    parts = [torch.empty(per, device=flat.device, dtype=torch.float32) for _ in range(world)]
    dist.all_gather(parts, mine)
    full = torch.cat(parts)[:n]
    off = 0
    for t in tensors:
        t.copy_(full[off:off + t.numel()].reshape(t.shape).to(device=t.device, dtype=t.dtype))
        off += t.numel()
    for t in others:
        dist.broadcast(t, src=src)


def load_on_rank0_and_broadcast(trainer, milestone, src=0):
    """north_star's start-up: ONLY rank ``src`` reads the checkpoint from disk (``trainer.load``: online weights + EMA
    copy, DDP:1804-1826); every other rank receives both copies over RCCL — one scatter + all-gather payload each
    (``broadcast_module_``) — instead of N processes each loading the file (README:14, DGM/dgm_sample.py:54), and the
    host-side counters a resumed ``Trainer.train`` loops on (``trainer.step``, the EMA's ``step`` / ``initted``, the
    optimiser state) as one object broadcast, so every rank runs the same number of steps.  A failing load on ``src``
    raises on EVERY rank (the peers would otherwise sit in the next collective until it times out).
    ``milestone`` None: nothing to load, the seeded initialisation of rank ``src`` is broadcast.  -> True if a file was read."""
    rank = dist.get_rank() if world_size() > 1 else 0
    loaded, err = False, None
    if rank == src and milestone is not None:
        try:
            trainer.load(milestone)
            loaded = True
        except Exception as e:                               # reported to every rank below, then re-raised
            err = e
            if world_size() == 1:
                raise
    if world_size() > 1:
        dev = next(trainer.model.parameters()).device
        # the EMA copy gets storage of its own only when the checkpoint's EMA weights differ from the online ones: every
        # rank has to make the same choice before the payloads travel
        flag = torch.tensor([int(trainer.ema.ema_model is not trainer.ema.online_model), int(loaded), int(err is not None)],
                            device=dev)
        dist.broadcast(flag, src=src)
        if bool(flag[2].item()):
            raise RuntimeError(f'rank {src} could not load checkpoint {milestone!r}' + (f': {err!r}' if err else ''))
        if bool(flag[0].item()) and trainer.ema.ema_model is trainer.ema.online_model:
            trainer.ema._own_copy()
        broadcast_module_(trainer.model, src=src)
        if trainer.ema.ema_model is not trainer.ema.online_model:
            broadcast_module_(trainer.ema.ema_model, src=src)
            trainer.ema._bump()
        loaded = bool(flag[1].item())
        if loaded:
            opt = trainer._ts.state_dict() if trainer._ts is not None else trainer._opt_state
            host = [dict(step=int(trainer.step), ema_step=int(trainer.ema.step), ema_initted=bool(trainer.ema.initted),
                         opt=_to_cpu(opt))] if rank == src else [None]
            dist.broadcast_object_list(host, src=src)
            if rank != src:
                trainer.step = host[0]['step']
                trainer.ema.step.fill_(host[0]['ema_step'])
                trainer.ema.initted.fill_(host[0]['ema_initted'])
                if host[0]['opt'] is not None:
                    trainer._opt_state = host[0]['opt']
                    if trainer._ts is not None:
                        trainer._ts.load_state_dict(trainer._opt_state)
    return loaded


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def average_gradients(grads, scale):
    """training: what accelerate's DDP wrapper does for the reference at DDP:1850 — every rank ends
    with the mean over ranks of each gradient.  ``grads``: {name: tensor}, same names and shapes on every rank;
    ``scale(flat, s)`` -> flat * s (a HIP kernel on the GPU path).  One bucket: the ~36 M gradient floats of the DGM UNet
    travel as a single all-reduce (RCCL picks its ring / direct algorithm over the xGMI mesh), issued after the
    backward pass — 1-2 ms against a ~40 ms step, so nothing is gained by slicing it under the backward kernels."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return grads
    names = sorted(grads)                                            # the same order on every rank
    flat = torch.cat([grads[k].reshape(-1) for k in names])
    dist.all_reduce(flat)
    flat = scale(flat, 1.0 / dist.get_world_size())
    out, o = {}, 0
    for k in names:
        n = grads[k].numel()
        out[k] = flat[o:o + n].view(grads[k].shape)
        o += n
    return out


def gather_records(imgs_u8, homos, dst=0):
    """per-rank {"imgs": uint8 (b,6,H,W), "homos": f64 (b,3,3)} tensors -> on ``dst`` the rank-ordered
    concatenation (the record of saveTrainPair, DDP:1678); None elsewhere.  Shards may differ in size
    (shard_bounds hands the first ``total % world`` ranks one sample more): the sizes are exchanged first and the
    short shards travel padded to the longest."""
    if world_size() == 1:
        return imgs_u8, homos
    world, rank = dist.get_world_size(), dist.get_rank()
    dev_out = imgs_u8.device
    imgs_u8, homos = _to_coll(imgs_u8), _to_coll(homos)
    nb = torch.tensor([imgs_u8.shape[0]], device=imgs_u8.device, dtype=torch.int64)
    sizes = [torch.empty_like(nb) for _ in range(world)]
    dist.all_gather(sizes, nb)
    sizes = [int(s.item()) for s in sizes]
    assert homos.shape[0] == imgs_u8.shape[0]
    bmax = max(sizes)

    def padded(t):
        if t.shape[0] == bmax:
            return t.contiguous()
        out = torch.zeros((bmax,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        out[:t.shape[0]] = t
        return out
    pi, ph = padded(imgs_u8), padded(homos)
    out_i = [torch.empty_like(pi) for _ in range(world)] if rank == dst else None
    out_h = [torch.empty_like(ph) for _ in range(world)] if rank == dst else None
    dist.gather(pi, out_i, dst=dst)
    dist.gather(ph, out_h, dst=dst)
    if rank != dst:
        return None, None
    return (torch.cat([t[:n] for t, n in zip(out_i, sizes)]).to(dev_out),
            torch.cat([t[:n] for t, n in zip(out_h, sizes)]).to(dev_out))


def noise_key(seed, total, rank, world, first_id=0):
    """(seed, global sample ids of ``rank``'s rows) — the key every rank derives for the sample-indexed generator
    (``cfg.DeviceRng.key_by_sample`` -> dmh_rng_indexed): the SAME seed everywhere, ids = first_id + this rank's contiguous
    slice of ``range(total)``.  The concatenation over ranks is ``first_id + range(total)``, so an N-rank run draws, row for
    row, what one process holding all ``total`` rows draws."""
    lo, hi = shard_bounds(total, rank, world)
    return int(seed), range(first_id + lo, first_id + hi)


def key_noise_by_sample(diffusion, seed, total, first_id=0, device=None):
    """key ``diffusion.rng`` for this process's shard of a global batch of ``total`` samples (rank / world from the process
    group; one process: the whole batch).  Replaces the reference's per-process stream generator (CFG:679,705,90; the N
    hand-started processes of DGM/dgm_sample.py:13-18 each draw from torch's default generator, seeded at random per process:
    row b of a draw depends on the shard and nothing is reproducible)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    seed, ids = noise_key(seed, total, rank, world, first_id)
    diffusion.rng.key_by_sample(seed, ids, device)
    return ids


def SampleIndexedRng(seed, sample_ids, device):
    """a generator keyed by global sample index (drop-in for a diffusion's ``.rng``): ``cfg.DeviceRng`` after
    ``key_by_sample`` — one dmh_rng_indexed launch per draw, graph-capturable."""
    from .cfg import DeviceRng
    return DeviceRng().key_by_sample(seed, sample_ids, device)

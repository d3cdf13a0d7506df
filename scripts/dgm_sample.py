#!/usr/bin/env python3
"""Drop-in counterpart of the reference's DGM/dgm_sample.py on dmhomo_amd (same CLI flags, same output format).

    python scripts/dgm_sample.py -c DGM --s_step 32 --bs 25 --exp run0 [--image_size 256] [--batches 2]

Differences from the reference script (DGM/dgm_sample.py:11-101), all forced by what is available offline:
  * conditions come from dmhomo_amd.ddpm.SyntheticConditions (the CA-Homo dataset of DDP:1058-1066 is not
    shipped) unless --conditions points at a .pt file holding an iterable of (12-channel batch, classes);
  * -c names results/model-<c>.pt like the reference; when the file does not exist the seeded random
    initialisation is used (the trained DGM.pt lives on HuggingFace, README:8);
  * the loop stops after --batches batches instead of running until killed (SAMPLE:62);
  * multi-GPU: launch with torch.distributed.run instead of N hand-started processes (--gpu_nums / -i are
    still accepted and select the data slice exactly as the reference's unused arguments did: not at all); rank 0
    alone reads the checkpoint and broadcasts the online + EMA weights over RCCL (the reference's N processes each
    load the file, SAMPLE:54); sampling replays one captured denoise step from a HIP graph, with the conditional pass's
    dropped rows not computed (cfg.Unet.dedup_dropped_rows: the same samples bit for bit); noise is keyed by --seed and
    the global sample index (dmh_rng_indexed).  With the synthetic conditions sample g of the job is also BUILT from
    g (ddpm.SyntheticConditions), so N processes write, between them, exactly the records one process writes in N times
    as many batches (tests/test_gpu_distributed.py).  With a dataset folder the loader deals rows to ranks in strides
    (dataset.ConditionLoader, as accelerate's sharded DataLoader does) while noise ids are contiguous per rank: the
    records are reproducible for a given N, but which noise meets which image depends on N.
Output: traindata/<exp>/dataset/idx_<i>_rank_<r>_part_<p>_dm_cahomo_<k>k.npy — a pickled list of
{"imgs": uint8 (B,6,H,W), "homos": float64 (B,3,3)} every 2 batches (SAMPLE:73-77), the format
HEM/dataset/data_loader.py:123-131 consumes.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmhomo_amd.denoising_diffusion_models.denoising_diffusion_pytorch import Trainer  # noqa: E402
from dmhomo_amd.denoising_diffusion_models.classifier_free_guidance import Unet, GaussianDiffusion  # noqa: E402
from dmhomo_amd import distributed as D  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument('-c', type=str, default='None')
parser.add_argument('--gpu_nums', type=int, default=0)
parser.add_argument('--s_step', type=int, default=0)
parser.add_argument('--part', type=int, default=0)
parser.add_argument('--bs', type=int, default=80)
parser.add_argument('--exp', type=str, default='exp')
parser.add_argument('-i', type=int, default=0)
parser.add_argument('--image_size', type=int, default=256)        # SAMPLE:32 hard-codes 256
parser.add_argument('--batches', type=int, default=2)
parser.add_argument('--conditions', type=str, default=None)
parser.add_argument('--seed', type=int, default=0, help='noise seed (every value is keyed by seed and global sample index)')
args = parser.parse_args()

num_classes = 1


def main():
    rank, world, device = D.init_from_env()
    torch.manual_seed(args.seed)         # (the initialisation used when no checkpoint is found; torch seeds its CPU generator at random)
    model = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=num_classes).to(device)
    model.cfg_mode = 'streams'
    diffusion = GaussianDiffusion(model, image_size=args.image_size, timesteps=1000, sampling_timesteps=args.s_step,
                                  loss_type='l1', objective='pred_x0').to(device)
    folder = torch.load(args.conditions) if args.conditions else 'DGM_Conditions'
    trainer = Trainer(diffusion, folder, train_batch_size=args.bs, train_lr=1e-4, train_num_steps=200000,
                      gradient_accumulate_every=2, ema_decay=0.995, amp=False, results_folder='results',
                      save_and_sample_every=2000, num_samples=4, augment_horizontal_flip=False, num_worker=0,
                      total_data_slice_idx=args.gpu_nums, data_slice_idx=args.i, shuffle=False,
                      split_batches=False)     # --bs is PER PROCESS, as for the reference's hand-started processes (SAMPLE:13-18)
    # rank 0 alone reads the checkpoint; the online and the EMA copy reach the other ranks as one RCCL payload each
    have = os.path.exists(os.path.join('results', f'model-{args.c}.pt')) if rank == 0 else None
    if not D.load_on_rank0_and_broadcast(trainer, args.c if have else None):
        if rank == 0:
            print(f'results/model-{args.c}.pt not found: sampling from the seeded random initialisation')
    sampler = trainer.ema.ema_model                        # what Trainer.sample draws from (DDP:1960)
    sampler.model.cfg_mode = 'streams'
    sampler.hip_graph = True                               # one captured denoise step replayed s_step times
    sampler.model.dedup_dropped_rows = True                # CFG:404,415-425: dropped conditional rows == null rows, not computed
    out_dir = f'traindata/{args.exp}/dataset/'
    os.makedirs(out_dir, exist_ok=True)
    train_list, part = [], args.part
    for b in range(args.batches):
        # noise keyed by (seed, GLOBAL sample index): rank r of N draws rows [b*bs*N + r*bs, +bs) of the job's noise, so
        # the noise of a sample does not depend on how many processes made the job (the reference's N hand-started
        # processes, SAMPLE:13-18, each draw from their own default generator: torch seeds it at random per process)
        # (a loader that ends on a short batch — dataset.ConditionLoader keeps it, like the reference's DataLoader — draws
        #  the first rows' ids: cfg.DeviceRng.ids_for)
        D.key_noise_by_sample(sampler, args.seed, args.bs * world, first_id=b * args.bs * world, device=device)
        ret = trainer.sample(args.i, device, step=len(train_list))
        train_list.append(ret)
        print(f'length of trainList {len(train_list)}')
        if len(train_list) % 2 == 0:
            np.save(f'{out_dir}idx_{args.i}_rank_{rank}_part_{part}_dm_cahomo_{len(train_list) * args.bs / 1000}k.npy',
                    np.array(train_list, dtype=object), allow_pickle=True)
            train_list.clear()
            part += 1
    # (captured denoise steps live in an LRU keyed by batch shape: a loader whose epochs end on a short batch captures twice per
    #  job, not twice per epoch)
    print(f'rank {rank}: graph captures {sampler.graph_captures}')


if __name__ == '__main__':
    main()

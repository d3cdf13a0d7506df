#!/usr/bin/env python3
"""Drop-in counterpart of the reference's training entry DGM/demo.py on dmhomo_amd.

    python scripts/demo.py [-c <milestone>] [--data <CA-Homo Train dir>] [--steps N] [--image_size 256] [--bs 128]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/demo.py ...

Same model / diffusion / Trainer arguments as DEMO:15-58 (Unet dim 64, mults (1,2,4,8), 6 channels, 5 classes; l1,
pred_x0, 1000 steps; batch 128, lr 5e-4, accumulate 1, EMA 0.995).  Differences, forced by what is available offline:
  * ``accelerate launch`` becomes torch.distributed.run: one process per GPU, the global batch split over the ranks
    (accelerate's split_batches=True, DDP:1717), gradients averaged with one RCCL all-reduce per step;
  * --data names the dataset directory (the reference hard-codes it, DDP:1058); without it the seeded synthetic
    conditions of dmhomo_amd.ddpm.SyntheticConditions are used;
  * --steps overrides the 112 500 steps of DEMO:34-45 so that a smoke run ends.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmhomo_amd.denoising_diffusion_models.denoising_diffusion_pytorch import Trainer  # noqa: E402
from dmhomo_amd.denoising_diffusion_models.classifier_free_guidance import Unet, GaussianDiffusion  # noqa: E402
from dmhomo_amd import distributed as D  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument('-c', type=int, default=0)
parser.add_argument('--data', type=str, default='DMHomo')
parser.add_argument('--steps', type=int, default=0)
parser.add_argument('--image_size', type=int, default=256)          # DEMO:24
parser.add_argument('--bs', type=int, default=256 // 2)             # DEMO:34
parser.add_argument('--results', type=str, default='results')
args = parser.parse_args()

num_classes = 5


def main():
    rank, world, device = D.init_from_env()
    model = Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=num_classes).to(device)
    diffusion = GaussianDiffusion(model, image_size=args.image_size, timesteps=1000, sampling_timesteps=32, loss_type='l1',
                                  objective='pred_x0').to(device)
    D.broadcast_module_(diffusion, src=0)
    train_batch_size = args.bs
    data_num, epoch = 450000, 32
    total = (data_num * epoch) // train_batch_size
    if rank == 0:
        print(f'total steps are: {total}')
    trainer = Trainer(diffusion, args.data, train_batch_size=train_batch_size, train_lr=1e-4 * 10 / 2,
                      train_num_steps=args.steps or total, gradient_accumulate_every=1, ema_decay=0.995, amp=False,
                      results_folder=args.results, save_and_sample_every=1000, num_samples=9,
                      augment_horizontal_flip=False)
    if args.c != 0:
        trainer.load(args.c)
    trainer.train(log=(lambda s, l: print(f'step {s}: loss: {float(l):.4f}', flush=True)) if rank == 0 else None)


if __name__ == '__main__':
    main()

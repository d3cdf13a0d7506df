"""dev tool: every dmh_conv2d launch of ONE denoise step (cond + null rows in one launch sequence, eager, HIP events around each
launch: exclusive durations) grouped by shape — launches per step, average us, algorithmic TFLOP/s, share of the step's conv time:
    python tools/step_conv_table.py [--bs 25] [--image_size 128] [--dim 64]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmhomo_amd import cfg, ddpm, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=25)
    ap.add_argument('--image_size', type=int, default=128)
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--steps', type=int, default=6, help='denoise steps to average over')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = cfg.Unet(dim=a.dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    model.cfg_mode = 'batched'
    diff = cfg.GaussianDiffusion(model, image_size=a.image_size, timesteps=1000, sampling_timesteps=a.steps, loss_type='l1',
                                 objective='pred_x0').to(dev)
    diff.hip_graph = False
    data, classes = next(ddpm.SyntheticConditions(a.image_size, a.bs, seed=1000, device=dev))
    rgb_flow, flow, mask = data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous()
    diff.sample(classes, rgb_flow, flow, mask)
    torch.cuda.synchronize()
    ops.CONV_LOG = []
    diff.sample(classes, rgb_flow, flow, mask)
    torch.cuda.synchronize()
    log, ops.CONV_LOG = ops.CONV_LOG, None
    groups = collections.OrderedDict()
    for e0, e1, k, stride, B, ho, wo, cin, cout, up, pro in log:
        key = (k, stride, up, cin, cout, ho, wo, B, pro)
        groups.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
    total = sum(sum(v) for v in groups.values())
    print(f'{len(log) // a.steps} conv launches per denoise step, {total / a.steps / 1e3:.2f} ms of them per step (B = {2 * a.bs} rows)')
    print(f'{"k":>2} {"s":>1} {"up":>2} {"cin":>4} {"cout":>4} {"out":>9} {"pro":>3} {"n/step":>6} {"avg us":>8} {"TFLOP/s":>8} {"share":>6}')
    for (k, stride, up, cin, cout, ho, wo, B, pro), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        fl = 2.0 * B * ho * wo * cin * cout * k * k
        print(f'{k:2d} {stride:1d} {up:2d} {cin:4d} {cout:4d} {ho:4d}x{wo:<4d} {int(pro):3d} {len(v) / a.steps:6.1f} {avg:8.1f} {fl / avg / 1e6:8.1f} '
              f'{100 * sum(v) / total:5.1f}%')


if __name__ == '__main__':
    main()

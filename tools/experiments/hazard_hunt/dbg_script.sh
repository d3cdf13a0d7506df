cd /root/repo
for i in 1 2; do rm -rf /tmp/w$i; mkdir -p /tmp/w$i; (cd /tmp/w$i && PYTHONPATH=/root/repo python /root/repo/scripts/dgm_sample.py -c absent --s_step 3 --bs 3 --exp run0 --image_size 128 --batches 2 --seed 5 > log.txt 2>&1); done
python - <<'PY'
import numpy as np, glob
a=np.load(glob.glob('/tmp/w1/traindata/run0/dataset/*.npy')[0],allow_pickle=True)
b=np.load(glob.glob('/tmp/w2/traindata/run0/dataset/*.npy')[0],allow_pickle=True)
for k in range(2):
    print('batch',k,'imgs equal',np.array_equal(a[k]['imgs'],b[k]['imgs']),'homos equal',np.array_equal(a[k]['homos'],b[k]['homos']), 'maxdiff', np.abs(a[k]['imgs'].astype(int)-b[k]['imgs'].astype(int)).max())
PY
python - <<'PY'
import sys; sys.path.insert(0,'/root/repo')
import torch
from dmhomo_amd.denoising_diffusion_models.classifier_free_guidance import Unet
m=Unet(dim=64, dim_mults=(1,2,4,8), channels=6, num_classes=1)
print('param digest', float(sum(p.double().sum() for p in m.parameters())), torch.initial_seed())
PY
python - <<'PY'
import sys; sys.path.insert(0,'/root/repo')
import torch
from dmhomo_amd.denoising_diffusion_models.classifier_free_guidance import Unet
m=Unet(dim=64, dim_mults=(1,2,4,8), channels=6, num_classes=1)
print('param digest', float(sum(p.double().sum() for p in m.parameters())), torch.initial_seed())
PY

# round 4: is the headline step sensitive to workgroup-count quantisation?  images/s per batch size, same box, same process
# environment (bs = 25 -> 200 / 400 workgroups per deep launch per stream; bs = 32 -> 256 / 512)
cd /root/repo
for bs in 25 32 25 32 28 24; do python bench.py --bs $bs --steps 4 --warmup 1 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bs', $bs, 'images/s %.2f' % d['value'], 'ms/denoise %.3f' % d['ms_per_denoise_step'], 'us per row-step %.2f' % (d['ms_per_denoise_step']*1e3/(2*$bs)))"; done

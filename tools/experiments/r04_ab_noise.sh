# round 4: the step's noise drawn on a side branch of the captured step (under the network) vs behind the join
cd /root/repo
python -m pytest tests/test_gpu_unet.py tests/test_gpu_rng.py -m gpu -q -k "graph or sharded or keyed or variants" 2>&1 | tail -2
for i in 1 2 3; do DMH_NOISE_UNDER_NETWORK=0 python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('serial', json.loads(sys.stdin.read())['value'])"; python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('under  ', json.loads(sys.stdin.read())['value'])"; done

# round 4: (scale, shift) rows of the replayed step from tables (one launch per pass) vs two embeddings + five linears per pass
cd /root/repo
python -m pytest tests/test_gpu_unet.py tests/test_gpu_rng.py tests/test_gpu_distributed.py -m gpu -q -k "graph or sharded or keyed or variants or stress or two_ranks" 2>&1 | tail -3
for i in 1 2 3 4; do DMH_SS_TABLES=0 python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('per-step', json.loads(sys.stdin.read())['value'])"; python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('tables  ', json.loads(sys.stdin.read())['value'])"; done

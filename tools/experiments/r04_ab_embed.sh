# round 4: embeddings + (scale, shift) rows of the two CFG passes computed beside the stem / first conv (side streams
# started early) vs at the head of each pass's trunk
cd /root/repo
python -m pytest tests/test_gpu_unet.py tests/test_gpu_rng.py -m gpu -q -k "graph or sharded or keyed or variants or modes_agree or bs25" 2>&1 | tail -2
for i in 1 2 3 4; do DMH_EMBED_EARLY=0 python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('late ', json.loads(sys.stdin.read())['value'])"; python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('early', json.loads(sys.stdin.read())['value'])"; done

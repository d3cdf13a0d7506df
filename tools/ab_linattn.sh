# A/B of two library builds on the fused LinearAttention shapes + the whole step: bash tools/ab_linattn.sh <prev.so>
PREV=$1
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "linattn or linear_attention or attention" 2>&1 | tail -1
for r in 1 2; do
  echo "== prev ($r)"; DMH_LIB_PATH=$PREV python tools/linattn_bench.py 2>&1 | grep fused
  echo "== new ($r)"; python tools/linattn_bench.py 2>&1 | grep fused
done
for i in 1 2 3; do DMH_LIB_PATH=$PREV python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('prev', json.loads(sys.stdin.read())['value'])"; python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('new', json.loads(sys.stdin.read())['value'])"; done

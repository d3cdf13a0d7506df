# A/B of a library build against dmhomo_amd/libdmhomo_prev.so (the previous build, same ABI version) on one box: kernel + soak tests,
# every conv shape, the fused LinearAttention shapes, then the whole step three times alternating (used for the six-instruction fp16 split).
#   gpurun -- bash tools/experiments/ab_split.sh
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_soak.py -m gpu -q -x 2>&1 | tail -2
bash tools/ab_conv.sh dmhomo_amd/libdmhomo_prev.so --bound 2>&1 | grep -E "==|us" | cut -c1-60 | awk '/==/{tag=$2 $3} /us/{print tag, $0}' | sort -k2,2 -s
for r in 1 2; do
  echo "== prev ($r)"; DMH_LIB_PATH=dmhomo_amd/libdmhomo_prev.so python tools/linattn_bench.py 2>&1 | grep fused
  echo "== new ($r)"; python tools/linattn_bench.py 2>&1 | grep fused
done
for i in 1 2 3; do DMH_LIB_PATH=dmhomo_amd/libdmhomo_prev.so python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('prev', json.loads(sys.stdin.read())['value'])"; python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('new', json.loads(sys.stdin.read())['value'])"; done

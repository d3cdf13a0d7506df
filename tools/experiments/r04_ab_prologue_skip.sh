# round 4, conv prologue: waves that hold no lane of the partly filled last load group skip its transcendentals
cd /root/repo
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv" 2>&1 | tail -2
bash tools/ab_conv.sh dmhomo_amd/libdmhomo_prev.so --reps 40 --only 3x3_64_64_128 2>&1 | grep -E "==|us" | cut -c1-70
bash tools/ab_conv.sh dmhomo_amd/libdmhomo_prev.so --reps 40 --only 3x3_128_128_64 2>&1 | grep -E "==|us" | cut -c1-70
for i in 1 2 3; do DMH_LIB_PATH=dmhomo_amd/libdmhomo_prev.so python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('prev', json.loads(sys.stdin.read())['value'])"; python bench.py --no-cpu-baseline --no-roofline --steps 4 2>/dev/null | python -c "import json,sys; print('new', json.loads(sys.stdin.read())['value'])"; done

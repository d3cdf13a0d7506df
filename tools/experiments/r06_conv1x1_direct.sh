# round 6: 1x1 convolutions with 64 output channels as independent one-wave workgroups without LDS staging (DMH_CONV1_DIRECT=1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c1; mkdir -p $O; cd $R
DMH_CONV1_DIRECT=1 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "conv" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
DMH_CONV1_DIRECT=1 python3 -m pytest tests/test_gpu_unet.py tests/test_gpu_dedup.py tests/test_gpu_parity_desat.py -q -x >> $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
grep -E "passed|failed|rc=|Error|assert" $O/tests.txt | tail -8
for r in 1 2 3; do
  for k in 0 1; do
    for B in 25 50; do
      echo "== DMH_CONV1_DIRECT=$k B=$B ($r)"
      DMH_CONV1_DIRECT=$k python3 tools/conv_bench.py --batch $B --only 1x1_128_64_128 --reps 40
      DMH_CONV1_DIRECT=$k python3 tools/conv_bench.py --batch $B --only 1x1_64+64_64_128 --reps 40
    done
  done
done > $O/conv_bench.txt 2>&1
grep -v amdgpu.ids $O/conv_bench.txt
for r in 1 2 3 4; do
  for k in 0 1; do
    echo "== DMH_CONV1_DIRECT=$k ($r)"
    DMH_CONV1_DIRECT=$k python3 bench.py --steps 6 --warmup 2 --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases | cut -c1-160
  done
done > $O/step.txt 2>&1
grep -v amdgpu.ids $O/step.txt | cut -c1-200

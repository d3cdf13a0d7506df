# round 6: pass 2 of the fully fused LinearAttention requests BOTH chunks of a sub-tile a sub-tile ahead (DMH_LA_PF2=1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6pf2; mkdir -p $O; cd $R
DMH_LA_PF2=1 python3 -m pytest tests/test_gpu_kernels.py -q -x -k "linear_attention" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
DMH_LA_PF2=1 python3 -m pytest tests/test_gpu_soak.py -q -x >> $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
for r in 1 2 3; do
  for k in 0 1; do
    echo "== DMH_LA_PF2=$k ($r)"
    DMH_LA_PF2=$k python3 tools/linattn_bench.py --only 0
    DMH_LA_PF2=$k python3 tools/linattn_bench.py --only 0 --rows 25
    DMH_LA_PF2=$k python3 tools/linattn_bench.py --only 1
  done
done > $O/linattn_bench.txt 2>&1
grep -v amdgpu.ids $O/linattn_bench.txt
DMH_LA_PF2=0 python3 tools/la_stamps.py > $O/la_stamps_0.txt 2>&1; DMH_LA_PF2=1 python3 tools/la_stamps.py > $O/la_stamps_1.txt 2>&1
grep -v amdgpu.ids $O/la_stamps_0.txt $O/la_stamps_1.txt
for r in 1 2 3 4; do
  for k in 0 1; do
    echo "== DMH_LA_PF2=$k ($r)"
    DMH_LA_PF2=$k python3 bench.py --steps 6 --warmup 2 --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases | cut -c1-160
  done
done > $O/step.txt 2>&1
grep -v amdgpu.ids $O/step.txt | cut -c1-200

# round 6, VERDICT r5 item 2: LinearAttention pass 1 hands its staged operand to pass 2.  output gpurun_out/r6la/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6la; mkdir -p $O; cd $R
python3 -m pytest tests/test_gpu_kernels.py -q -x -k "linear_attention" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
python3 -m pytest tests/test_gpu_soak.py tests/test_isa_hazards.py -q -x >> $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
for r in 1 2 3; do
  for k in 0 1; do
    echo "== DMH_LA_PIECES=$k ($r)"
    DMH_LA_PIECES=$k python3 tools/linattn_bench.py --only 0
    DMH_LA_PIECES=$k python3 tools/linattn_bench.py --only 0 --rows 25
    DMH_LA_PIECES=$k python3 tools/linattn_bench.py --only 1
  done
done > $O/linattn_bench.txt 2>&1
grep -v amdgpu.ids $O/linattn_bench.txt
# per-kernel times of the two passes (kernel trace of the bench tool)
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  DMH_LA_PIECES=$k rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$k -o t -- python3 $R/tools/linattn_bench.py --only 0 > $O/trace$k.log 2>&1
  echo "== DMH_LA_PIECES=$k kernel stats (C=64 128x128 B=50)"; f=$(find $O/trace$k -name '*kernel_stats.csv' | head -1); grep -E "linattn|pixel_stats" $f | cut -d, -f1-4 | cut -c1-150
done > $O/kernel_stats.txt 2>&1
cat $O/kernel_stats.txt
rm -rf $O/trace0 $O/trace1
cd $R
DMH_LA_PIECES=0 python3 tools/la_stamps.py > $O/la_stamps_own.txt 2>&1; DMH_LA_PIECES=1 python3 tools/la_stamps.py > $O/la_stamps_pieces.txt 2>&1
grep -v amdgpu.ids $O/la_stamps_own.txt $O/la_stamps_pieces.txt
for r in 1 2 3; do
  for k in 0 1; do
    echo "== DMH_LA_PIECES=$k ($r)"
    DMH_LA_PIECES=$k python3 bench.py --steps 6 --warmup 2 --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases | cut -c1-160
    DMH_LA_PIECES=$k python3 bench.py --steps 6 --warmup 2 --dedup --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases | cut -c1-200
  done
done > $O/step.txt 2>&1
grep -v amdgpu.ids $O/step.txt | cut -c1-200
# the de-duplicating step as ONE launch sequence (cond + null rows listed together) instead of two streams
python3 bench.py --steps 6 --warmup 2 --dedup --cfg-mode batched --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases | cut -c1-200 > $O/dedup_batched.txt 2>&1
cat $O/dedup_batched.txt

# round 6, VERDICT r5 item 1: the <= 32^2 convolutions.  gpurun -- bash tools/experiments/r06_deep_levels.sh ; output gpurun_out/r6deep/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6deep; mkdir -p $O; cd $R
# 1. where the conv time of a denoise step goes, per shape (batched 50 rows; 24 rows = one CFG pass's launch sizes)
python3 tools/step_conv_table.py > $O/step_conv_table_b50.txt 2>&1
python3 tools/step_conv_table.py --bs 12 > $O/step_conv_table_b24.txt 2>&1
# 2. split-K emulated exactly with the product kernel (Cin / ks channels on B * ks rows) + the reduce's HBM floor
python3 tools/ksplit_probe.py > $O/ksplit_probe.txt 2>&1
# 3. cout tiles of one pixel tile on the SAME XCD at the deep levels: time (alternating), then traffic
for r in 1 2; do
  for B in 25 50; do
    for k in 0 32; do
      echo "== DMH_CONV_XCD_DEEP=$k B=$B ($r)"
      DMH_CONV_XCD_DEEP=$k python3 tools/conv_bench.py --batch $B --only 3x3_512_512_16 --reps 40
      DMH_CONV_XCD_DEEP=$k python3 tools/conv_bench.py --batch $B --only 3x3_256_256_32 --reps 40
    done
  done
done > $O/xcd_deep_time.txt 2>&1
for k in 0 32; do
  for s in 3x3_512_512_16 3x3_256_256_32; do
    echo "== DMH_CONV_XCD_DEEP=$k $s: FETCH_SIZE (KiB, x2 on gfx950) / WRITE_SIZE per launch"
    DMH_CONV_XCD_DEEP=$k PMC=FETCH_SIZE bash tools/pmc_conv.sh $s
    DMH_CONV_XCD_DEEP=$k PMC=WRITE_SIZE bash tools/pmc_conv.sh $s
  done
done > $O/xcd_deep_traffic.txt 2>&1
cd $R
# 4. the step, alternating (headline, then the de-duplicating variant as the timed loop)
for r in 1 2 3; do
  for k in 0 32; do
    echo "== DMH_CONV_XCD_DEEP=$k ($r)"
    DMH_CONV_XCD_DEEP=$k python3 bench.py --steps 6 --warmup 2 --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases | cut -c1-160
    DMH_CONV_XCD_DEEP=$k python3 bench.py --steps 6 --warmup 2 --dedup --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases | cut -c1-200
  done
done > $O/xcd_deep_step.txt 2>&1
tail -3 $O/*.txt

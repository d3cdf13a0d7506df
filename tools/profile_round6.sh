# rocprofv3 passes behind profiles/r06_* (round 6).  gpurun -- bash tools/profile_round6.sh ; outputs under gpurun_out/r6/,
# copied into profiles/ by hand.  Every counter pass is its own run with --kernel-trace only (no other trace domain).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6
rm -rf $O && mkdir -p $O
stats() { find $1 -name '*kernel_stats.csv' | head -1; }
# 1. headline line, streams / batched kernel traces, PMC traffic of the 3x3 launches
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/streams -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-variants --no-cpu-baseline --no-roofline --no-phases > $O/streams.log 2>&1
cp "$(stats $O/streams)" $O/bench_streams_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/batched -o b -- python3 $R/bench.py --steps 2 --warmup 1 --cfg-mode batched --no-variants --no-cpu-baseline --no-roofline --no-phases > $O/batched.log 2>&1
cp "$(stats $O/batched)" $O/bench_batched_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --s_step 2 --cfg-mode batched --no-variants --no-cpu-baseline --no-roofline --no-phases --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --s_step 2 --cfg-mode batched --no-variants --no-cpu-baseline --no-roofline --no-phases --no-graph > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 'conv_f16x3_kernel<3, 3, 1, 0' $O/pmc_traffic.json > /dev/null
# (the same comparison bench.py prints as roofline.traffic_vs_committed: a fresh pass against the NEWEST committed figure)
python3 $R/tools/check_traffic.py $O/pmc_traffic.json $R/profiles > $O/traffic_check.txt 2>&1; echo "traffic check exit $?" >> $O/traffic_check.txt
# 1b. the de-duplicating step (cfg.Unet.dedup_dropped_rows) as the timed loop: kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dedup -o s -- python3 $R/bench.py --steps 2 --warmup 1 --dedup --no-variants --no-cpu-baseline --no-roofline --no-phases > $O/dedup.log 2>&1
cp "$(stats $O/dedup)" $O/bench_dedup_streams_kernel_stats.csv
if [ "$FULL" = "1" ]; then
# 2. the README geometry (dim 64, 256x256, bs 25, s_step 32) and BASELINE configs[4] (dim 128, 256x256, bs 8, s_step 250): NOT the headline
python3 $R/bench.py --dim 64 --image_size 256 --bs 25 --s_step 32 --steps 2 --warmup 1 > $O/bench_256.json 2> $O/bench_256.err
python3 $R/bench.py --dim 128 --image_size 256 --bs 8 --s_step 250 --steps 1 --warmup 1 > $O/bench_stress.json 2> $O/bench_stress.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stress -o t -- python3 $R/bench.py --dim 128 --image_size 256 --bs 8 --s_step 16 --steps 1 --warmup 1 --cfg-mode batched --no-cpu-baseline --no-roofline > $O/stress.log 2>&1
cp "$(stats $O/stress)" $O/bench_stress_batched_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/g256 -o t -- python3 $R/bench.py --dim 64 --image_size 256 --bs 25 --s_step 8 --steps 1 --warmup 1 --cfg-mode batched --no-cpu-baseline --no-roofline > $O/g256.log 2>&1
cp "$(stats $O/g256)" $O/bench_256_batched_kernel_stats.csv
# 3. training step (BASELINE configs[3], one GPU's share)
python3 $R/bench.py --workload train --steps 5 --warmup 2 > $O/train_bench.json 2> $O/train_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python3 $R/tools/train_bench.py --steps 3 --warmup 1 > $O/train.log 2>&1
cp "$(stats $O/train)" $O/train_kernel_stats.csv
fi
# 4. the fused LinearAttention passes in isolation on HBM-resident inputs: counters, phase stamps (needs `make stamps`)
rm -rf $O/streams $O/batched $O/dedup $O/pmc_fetch $O/pmc_write $O/stress $O/g256 $O/train
# 3b. the canonical launch's counters (tools/pmc_round.sh writes gpurun_out/pmc_round/pmc_canonical.json)
bash $R/tools/pmc_round.sh > $O/pmc_round.log 2>&1; cp $R/gpurun_out/pmc_round/pmc_canonical.json $O/pmc_canonical.json
bash $R/tools/pmc_linattn.sh > $O/pmc_linattn.log 2>&1
cp $R/gpurun_out/pmc_la/pmc_linattn.json $O/pmc_linattn.json
if [ -f $R/dmhomo_amd/libdmhomo_hip_stamps.so ]; then
  (cd $R && python3 tools/kv_stamps.py 50 && python3 tools/la_stamps.py) > $O/linattn_stamps.txt 2>&1
fi
# 5. power / clock sample under the headline bench
(cd $R && bash tools/power_sample.sh) > $O/power_sample.log 2>&1; cp $R/gpurun_out/power_sample.txt $O/power_sample.txt
grep -h '"metric"' $O/bench.json $O/bench_256.json $O/bench_stress.json $O/train_bench.json 2>/dev/null | cut -c1-220
ls -la $O

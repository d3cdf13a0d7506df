# rocprofv3 passes behind profiles/ (round 2): kernel stats of the default and the batched bench command, the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) over the 3x3 launches of a shortened batched run, the
# ablation of the product 3x3 kernel, and the headline JSON line itself.
#   gpurun -- bash tools/profile_round.sh      (outputs under gpurun_out/round/, copied into profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
rm -rf $O && mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/streams -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/streams.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/batched -o b -- python3 $R/bench.py --steps 2 --warmup 1 --cfg-mode batched --no-cpu-baseline --no-roofline > $O/batched.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --s_step 2 --cfg-mode batched --no-cpu-baseline --no-roofline --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --s_step 2 --cfg-mode batched --no-cpu-baseline --no-roofline --no-graph > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 'conv_f16x3_kernel<3, 3, 1, 0' $O/pmc_traffic.json
if [ -f $R/dmhomo_amd/libdmhomo_hip_stamps.so ]; then
  python3 $R/tools/f16_ablate.py > $O/f16_ablate.log 2>&1
  python3 $R/tools/f16_stamps.py > $O/f16_stamps.log 2>&1
fi
grep -h '"metric"' $O/bench.json $O/streams.log $O/batched.log | cut -c1-200
ls -R $O | head -40

# rocprofv3 passes behind profiles/: kernel stats of the default and the batched bench command, and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) on a shortened batched run.
#   gpurun -- bash tools/profile_round.sh      (outputs under gpurun_out/round/, copied into profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/streams -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/streams.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/batched -o b -- python3 $R/bench.py --steps 2 --warmup 1 --cfg-mode batched --no-cpu-baseline > $O/batched.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --s_step 2 --cfg-mode batched --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --s_step 2 --cfg-mode batched --no-cpu-baseline > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 'conv_f16x3_kernel<3, 3' $O/pmc_traffic.json
grep -h '"metric"' $O/streams.log $O/batched.log | cut -c1-200
ls -R $O | head -40

# rocprofv3 PMC evidence behind DESIGN.md §3.1 / bench.py's roofline, for the canonical fused conv3x3 launch
# (64->64 @128^2, B = 50 rows, GN+SiLU prologue, GroupNorm partials).  Every pass is its own run with --kernel-trace only
# (MI355X_MICROARCH.md: counters in separate passes; FETCH_SIZE and WRITE_SIZE do not fit one pass); the program sits
# directly behind `--`.      gpurun -- bash tools/pmc_round.sh ; then  cp gpurun_out/pmc_round/r0N_pmc_canonical.json profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_round
rm -rf $O && mkdir -p $O
B="python3 $R/tools/conv_bench.py --only 3x3_64_64_128 --reps 10"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_random -o p -- $B > $O/sq_random.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq_zeros -o p -- $B --zeros > $O/sq_zeros.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq_mops -o p -- $B > $O/sq_mops.log 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE --output-format csv -d $O/ta -o p -- $B > $O/ta.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/cache -o p -- $B > $O/cache.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/time_random -o p -- $B > $O/time_random.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/time_zeros -o p -- $B --zeros > $O/time_zeros.log 2>&1
python3 $R/tools/pmc_round.py $O $O/pmc_canonical.json

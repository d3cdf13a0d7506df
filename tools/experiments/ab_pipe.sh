# A/B of tools/experiments/conv_f16x3_pipelined_prologue.patch (apply it, rebuild): conv parity cases, then tools/conv_bench.py --bound
# with DMH_CONV_PIPE=0 / 1 alternating on one box.   gpurun -- bash tools/experiments/ab_pipe.sh
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -4
for i in 1 2; do
for pp in 0 1; do
echo "== DMH_CONV_PIPE=$pp"
DMH_CONV_PIPE=$pp python tools/conv_bench.py --reps 30 --only 3x3 --bound 2>&1 | grep -v "^$" | head -12
done; done

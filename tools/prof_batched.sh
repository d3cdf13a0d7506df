cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_batched -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cfg-mode batched --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_batched.log 2>&1

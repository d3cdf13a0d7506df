# dev tool (round 4): where a denoise step of the headline command is NOT filling the chip — from a rocprofv3 kernel trace,
# per kernel family: the time during which ONLY launches of that family (and nothing big) were running.
#   gpurun -- bash tools/step_timeline.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl && mkdir -p $R/gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o g -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/tl/log.txt 2>&1
python3 $R/tools/step_timeline.py $(find $R/gpurun_out/tl -name '*kernel_trace.csv' | head -1)

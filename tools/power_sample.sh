# samples rocm-smi (power, clocks, temperature) every 0.5 s while the headline bench runs: bash tools/power_sample.sh
# (read-only queries; output gpurun_out/power_sample.txt)
mkdir -p gpurun_out
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showtemp --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_sample.txt &
SMI=$!
python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 2 2>/dev/null | cut -c1-160
kill $SMI 2>/dev/null
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
head -3 gpurun_out/power_sample.txt; echo ...; sed -n 20,26p gpurun_out/power_sample.txt

cd /root/repo
python tools/experiments/tiny_launch_bound.py 2>&1 | tail -14
for b in 25 32 50 64; do for w in 1 0; do echo "== batch $b WIDE3=$w"; DMH_F16_WIDE3=$w python tools/conv_bench.py --reps 30 --batch $b --only 3x3_512_512_16 2>&1 | tail -1; DMH_F16_WIDE3=$w python tools/conv_bench.py --reps 30 --batch $b --only 3x3_256_256_32 2>&1 | tail -1;  done; done

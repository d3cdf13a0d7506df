export DMH_CONV3_VARIANT=10
for a in 0 1 2 3 4 7 8 16 20 12 15 31; do echo "abl=$a"; DMH_WINOF_ABL=$a timeout 120 python tools/conv_bench.py --only 3x3_64_64_128 2>&1 | grep "^3x3_64_64_128 "; DMH_WINOF_ABL=$a timeout 120 python tools/conv_bench.py --only 3x3_512_512_16 2>&1 | grep "^3x3"; done

# A/B of two library builds on the same box: bash tools/ab_conv.sh <prev.so> [conv_bench args]
# (alternates the two builds so that clock / thermal drift hits both)
PREV=$1; shift
for r in 1 2; do
  echo "== prev ($r)"; DMH_LIB_PATH=$PREV python tools/conv_bench.py "$@"
  echo "== new ($r)"; python tools/conv_bench.py "$@"
done

# round 5: where the de-duplicating step's time goes.  gpurun -- bash tools/profile_dedup.sh ; outputs under gpurun_out/r5d/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5d
rm -rf $O && mkdir -p $O
stats() { find $1 -name '*kernel_stats.csv' | head -1; }
for mode in streams batched; do
  python3 $R/bench.py --steps 4 --warmup 1 --cfg-mode $mode --no-cpu-baseline --no-roofline > $O/bench_$mode.json 2> $O/bench_$mode.err
done
python3 $R/bench.py --steps 4 --warmup 1 --cfg-mode streams --stream-splits 2 --no-cpu-baseline --no-roofline > $O/bench_streams_split2.json 2>> $O/bench_streams.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dd -o s -- python3 $R/bench.py --steps 2 --warmup 1 --dedup --no-variants --no-cpu-baseline --no-roofline > $O/dedup_streams.log 2>&1
cp "$(stats $O/dd)" $O/bench_dedup_streams_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ff -o s -- python3 $R/bench.py --steps 2 --warmup 1 --no-variants --no-cpu-baseline --no-roofline > $O/full_streams.log 2>&1
cp "$(stats $O/ff)" $O/bench_full_streams_kernel_stats.csv
rm -rf $O/dd $O/ff
for f in $O/bench_*.json; do echo $f; python3 -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][0])
v=d.get('variants',{}).get('dedup_dropped_rows',{})
print(d['config']['cfg_mode'], 'value', round(d['value'],2), 'dedup', round(v.get('value',0),2), 'x', round(v.get('speedup_vs_value',0),3), 'rows', v.get('unet_rows_per_denoise_step'))
"; done

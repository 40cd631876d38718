mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t7.log 2>&1; rc=$?
tail -6 gpurun_out/t7.log
[ $rc -eq 0 ] || exit $rc
for g in 1 0; do
  SDE_GRAPH=$g timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b7_$g.json 2> gpurun_out/b7_$g.err || { tail -5 gpurun_out/b7_$g.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/b7_$g.json')); print('graph=$g', d['value'], d['unet_ms_per_call'], d['host_enqueue_ms_per_unet_call'])"
done
for g in 1 0; do
  SDE_GRAPH=$g timeout -k 10 200 python bench.py --steps 4 --warmup 2 --batch 1 --no-cpu-baseline > gpurun_out/b7_b1_$g.json 2> gpurun_out/b7_b1_$g.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/b7_b1_$g.json')); print('batch1 graph=$g', d['value'], d['unet_ms_per_call'], d['host_enqueue_ms_per_unet_call'])"
done

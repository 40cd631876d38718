mkdir -p gpurun_out
timeout -k 10 120 python -u -m pytest tests/wip_no_alloc.py -m gpu -x -q --timeout 100 --timeout-method thread > gpurun_out/t6.log 2>&1; rc=$?
tail -5 gpurun_out/t6.log
[ $rc -eq 0 ] || exit $rc
SDE_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 1 --warmup 1 --denoise-steps 4 --no-cpu-baseline > gpurun_out/b6_spawn.json 2> gpurun_out/b6_spawn.err || { tail -5 gpurun_out/b6_spawn.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b6_spawn.json')); print('spawn2', d['n_gpus'], d['value'], d['config']['parallelism'])"
SDE_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 1 --warmup 0 --denoise-steps 4 --prompts 6 --batch 4 --scheduler euler_a --no-cpu-baseline > gpurun_out/b6_strong.json 2> gpurun_out/b6_strong.err || { tail -5 gpurun_out/b6_strong.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/b6_strong.json')); print('strong', d['n_gpus'], d['value'], d['scaling'], d['config']['global_batch'])"

mkdir -p gpurun_out/c5
python -c "import torch" 2>/dev/null
( AB_TAGS="base final prof final_prof" bash profiles/ab_k1_residency.sh ) > gpurun_out/c5/ab.log 2>&1
( for tag in base final; do echo "== $tag small entries"; MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so timeout 60 python tests/perf_probe.py 512 200000 8192 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; done ) >> gpurun_out/c5/ab.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_lzma.py tests/test_gpu_inflate.py -x -q 2>&1 | grep -v "^E   \s*$" | cut -c1-700 | tail -30 ) > gpurun_out/c5/tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_streams.py -x -q -s 2>&1 | grep -v "^E   \s*$" | cut -c1-700 | tail -30 ) > gpurun_out/c5/streams.log 2>&1
( timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/c5/bench_cfg4.log 2> gpurun_out/c5/bench_cfg4.err
cat gpurun_out/c5/ab.log; tail -12 gpurun_out/c5/tests.log; tail -12 gpurun_out/c5/streams.log; python -c "
import json;j=json.loads(open('gpurun_out/c5/bench_cfg4.log').read().strip().splitlines()[-1]);print('cfg4',j['value'],j['ms_per_step'],j['config'].get('unique_streams'))"

mkdir -p gpurun_out/c3
python -c "import torch" 2>/dev/null
( AB_TAGS="base new walkfn prof walkfn_prof" bash profiles/ab_k1_residency.sh ) > gpurun_out/c3/ab.log 2>&1
( for tag in base walkfn; do echo "== $tag small entries"; MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so timeout 60 python tests/perf_probe.py 512 200000 8192 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; done ) >> gpurun_out/c3/ab.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_lzma.py tests/test_gpu_streams.py -x -q 2>&1 | tail -8 ) > gpurun_out/c3/tests.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_dropin.py -q 2>&1 | grep -v "^E   \s*$" | cut -c1-600 | tail -150 ) > gpurun_out/c3/dropin.log 2>&1
( timeout 300 python tests/fuzz_gpu.py 4000 7 2>&1 | tail -3 ) > gpurun_out/c3/fuzz.log 2>&1
( time timeout 1200 python bench.py ) > gpurun_out/c3/bench.log 2> gpurun_out/c3/bench.err
cat gpurun_out/c3/ab.log gpurun_out/c3/tests.log gpurun_out/c3/fuzz.log; tail -c 3000 gpurun_out/c3/bench.log; tail -5 gpurun_out/c3/bench.err; grep -n "AssertionError\|passed\|failed" gpurun_out/c3/dropin.log | cut -c1-700

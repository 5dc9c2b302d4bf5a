mkdir -p gpurun_out/c4
python -c "import torch" 2>/dev/null
( AB_TAGS="base walkfn emitfn emitfn_far2 emitfn_near2 prof emitfn_prof" bash profiles/ab_k1_residency.sh ) > gpurun_out/c4/ab.log 2>&1
( for tag in base emitfn emitfn_far2; do echo "== $tag small entries"; MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so timeout 60 python tests/perf_probe.py 512 200000 8192 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; done ) >> gpurun_out/c4/ab.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_dropin.py tests/test_gpu_streams.py tests/test_gpu_prime.py tests/test_gpu_prime_write.py -x -q 2>&1 | grep -v "^E   \s*$" | cut -c1-700 | tail -40 ) > gpurun_out/c4/tests.log 2>&1
( timeout 300 python tests/fuzz_gpu.py 4000 7 2>&1 | tail -3 ) > gpurun_out/c4/fuzz.log 2>&1
cat gpurun_out/c4/ab.log gpurun_out/c4/fuzz.log; tail -15 gpurun_out/c4/tests.log

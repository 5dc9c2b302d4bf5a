mkdir -p gpurun_out/c7
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests/test_gpu_hash.py tests/test_gpu_prime.py tests/test_gpu_archive.py tests/test_gpu_dropin.py -x -q 2>&1 | grep -v "^E   \s*$" | cut -c1-900 | tail -40 ) > gpurun_out/c7/tests.log 2>&1
tail -30 gpurun_out/c7/tests.log

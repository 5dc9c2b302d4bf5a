#!/bin/bash
# Round 4, call 22: method 95 WRITE with the dictionary kept across LZMA2 chunks, on the device: encoder tests, the drop-in
# stream / archive / CLI tests that write .xz, the .xz decoder's own tests
set -u
mkdir -p gpurun_out/c22
python -c "import torch" 2>/dev/null
( timeout 900 python -X faulthandler -m pytest tests/test_gpu_lzma_enc.py tests/test_gpu_xz.py tests/test_gpu_cli.py tests/test_gpu_dropin.py tests/test_gpu_prime_write.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/c22/test_xz_write.log 2>&1
cat gpurun_out/c22/*.log

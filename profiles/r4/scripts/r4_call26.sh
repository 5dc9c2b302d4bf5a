#!/bin/bash
# Round 4, call 26: what the last three commits touched, on the device once more (the segment path through the shared parse
# launcher, K3 with its default back to compare-then-subtract, the CLI's lzma / xz rows)
set -u
mkdir -p gpurun_out/c26
python -c "import torch" 2>/dev/null
( timeout 400 python -X faulthandler -m pytest tests/test_gpu_dropin.py tests/test_gpu_prime_write.py tests/test_gpu_lzma.py tests/test_gpu_xz.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
  timeout 200 python -X faulthandler -m pytest tests/test_gpu_cli.py -x -q -k "lzma or xz" 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/c26/tests.log 2>&1
cat gpurun_out/c26/tests.log

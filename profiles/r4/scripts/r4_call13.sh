#!/bin/bash
# Round 4, call 13: the prime / DeviceArchive tests of large entries again (the first had a bug in the test's child program)
set -u
mkdir -p gpurun_out/c13
python -c "import torch" 2>/dev/null
( timeout 600 python -m pytest tests/test_gpu_prime.py tests/test_gpu_archive.py -x -q -s -k "large" 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/c13/prime_archive.log 2>&1
cat gpurun_out/c13/*.log

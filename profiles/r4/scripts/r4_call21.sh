#!/bin/bash
# Round 4, call 21: the GPU tests behind the one that died in call 20 (its call passed 14 arguments to a function of 16)
set -u
mkdir -p gpurun_out/c21
python -c "import torch" 2>/dev/null
( timeout 1200 python -X faulthandler -m pytest tests/test_gpu_streams.py tests/test_gpu_wrappers.py tests/test_gpu_xz.py -v -x -s 2>&1 | grep -v amdgpu.ids ) > gpurun_out/c21/gputest_rest.log 2>&1
grep -n "PASSED\|FAILED\|ERROR\|Fatal\|passed\|failed\|one window\|GiB" gpurun_out/c21/gputest_rest.log | tail -40

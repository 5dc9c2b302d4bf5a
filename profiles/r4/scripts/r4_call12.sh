#!/bin/bash
# Round 4, call 12: large entries where they lie (mzhip_inflate_large: the prime, DeviceArchive), the wrappers in window
# mode and the window fuzz on the device for the first time
set -u
mkdir -p gpurun_out/c12
python -c "import torch" 2>/dev/null
( MZHIP_PAR_TRACE=0 timeout 600 python -m pytest tests/test_gpu_streams.py -x -q -s -k "large_entry_device" 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/c12/large_dev.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_prime.py tests/test_gpu_archive.py -x -q -s -k "large" 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/c12/prime_archive.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_wrappers.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -8 ) > gpurun_out/c12/wrappers.log 2>&1
( timeout 600 python tests/fuzz_gpu_windows.py 60 21 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/c12/fuzz_windows.log 2>&1
cat gpurun_out/c12/*.log

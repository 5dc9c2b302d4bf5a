#!/bin/bash
# Round 4, call 9: the large-entry path with the window in page-locked memory and the search bounded by the last window's
# ratio; the two WRITE_SIZE passes again (bench.py no longer terminates its worker pool under the profiler).
set -u
mkdir -p gpurun_out/c9
python -c "import torch" 2>/dev/null
( timeout 300 python -m pytest tests/test_gpu_streams.py -x -q -s -k "one_window" 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/c9/one_window.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -s -k "many_waves or window_mode" 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/c9/dropin.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_streams.py -x -q -s -k "larger_than_any" 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/c9/large.log 2>&1
cat gpurun_out/c9/one_window.log gpurun_out/c9/dropin.log gpurun_out/c9/large.log
MZ_COLLECT_TIMEOUT=150 bash profiles/collect.sh c9_w2 WRITE_SIZE > gpurun_out/c9/collect_w2.log 2>&1
MZ_COLLECT_CONFIG=5 MZ_COLLECT_KERNEL=k_deflate_batch MZ_COLLECT_TIMEOUT=150 bash profiles/collect.sh c9_w5 WRITE_SIZE > gpurun_out/c9/collect_w5.log 2>&1
ls -la gpurun_out/c9 gpurun_out/c9_w2 gpurun_out/c9_w5 | head -40

#!/bin/bash
# Round 4, call 11: the large-entry path after (a) window mode entered as soon as 4 MiB have been decoded, (b) in[] in
# page-locked memory, (c) four links per pointer-jump round
set -u
mkdir -p gpurun_out/c11
python -c "import torch" 2>/dev/null
( timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -s -k "many_waves or window_mode" 2>&1 | grep -v amdgpu.ids | tail -5 ) > gpurun_out/c11/dropin.log 2>&1
( timeout 300 python -m pytest tests/test_gpu_streams.py -x -q -s -k "one_window" 2>&1 | grep -v amdgpu.ids | tail -5 ) >> gpurun_out/c11/dropin.log 2>&1
for k in "text 1 1" "sparse 3 1" "mixed 1 6"; do
  ( MZ_PERF_SKIP_OFF=1 timeout 600 python tests/perf_large_entry.py $k 2>&1 | grep -v amdgpu.ids ) >> gpurun_out/c11/large_entry.log 2>&1
done
cat gpurun_out/c11/dropin.log; cut -c1-400 gpurun_out/c11/large_entry.log

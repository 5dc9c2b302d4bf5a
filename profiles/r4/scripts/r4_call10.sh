#!/bin/bash
# Round 4, call 10: where a large entry's time goes (the shim's and the device call's own clocks)
set -u
mkdir -p gpurun_out/c10
python -c "import torch" 2>/dev/null
for k in "text 1 1" "sparse 3 1" "text 0.5 6" "mixed 1 6"; do
  ( timeout 600 python tests/perf_large_entry.py $k 2>&1 | grep -v amdgpu.ids ) >> gpurun_out/c10/large_entry.log 2>&1
done
cat gpurun_out/c10/large_entry.log

#!/bin/bash
# Round 5, call 33: the two window-mode differential runs on the device, on the round's HEAD
set -u
root=$PWD; out=$root/gpurun_out/c33; mkdir -p $out
( timeout 300 python tests/fuzz_gpu_windows.py 90 12 2>&1 | grep -v amdgpu.ids | tail -2 ) > $out/fuzz_gpu_windows.log 2>&1
( timeout 300 python tests/fuzz_xz_windows.py 12 12 2>&1 | grep -v amdgpu.ids | tail -2 ) > $out/fuzz_xz_windows.log 2>&1
cat $out/*.log

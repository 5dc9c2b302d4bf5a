#!/bin/bash
# Round 5, call 20: the long differential runs on HEAD -- K1 against the oracle (60 000 streams), the window-mode fuzzes of the DEFLATE
# and the .xz READ streams against the all-reference build
set -u
root=$PWD; out=$root/gpurun_out/c20; mkdir -p $out
( timeout 900 python tests/fuzz_gpu.py 12000 7 2>&1 | grep -v amdgpu.ids | tail -3 ) > $out/fuzz_gpu.log 2>&1
( timeout 900 python tests/fuzz_xz_windows.py 24 5 2>&1 | grep -v amdgpu.ids | tail -5 ) > $out/fuzz_xz_windows.log 2>&1
( timeout 900 python tests/fuzz_gpu_windows.py 30 9 2>&1 | grep -v amdgpu.ids | tail -5 ) > $out/fuzz_gpu_windows.log 2>&1
cat $out/fuzz_gpu.log $out/fuzz_xz_windows.log $out/fuzz_gpu_windows.log

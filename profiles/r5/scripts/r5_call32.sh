#!/bin/bash
# Round 5, call 32: the review's gate on the round's HEAD -- tests/fuzz_gpu.py 12000 7 (60 000 streams through the C ABI against the oracle)
set -u
root=$PWD; out=$root/gpurun_out/c32; mkdir -p $out
( timeout 420 python tests/fuzz_gpu.py 12000 7 2>&1 | grep -v amdgpu.ids | tail -3 ) > $out/fuzz_gpu.log 2>&1
cat $out/fuzz_gpu.log

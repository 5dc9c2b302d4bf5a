#!/bin/bash
# Round 5, call 38: method 14 in windows decodes on behind TOTAL_OUT_MAX as the reference's read() does (a stream cut inside its end
# marker fails in the call that would have returned the entry's last bytes): the drop-in's LZMA tests on the device
set -u
root=$PWD; out=$root/gpurun_out/c38; mkdir -p $out
( timeout 300 python -m pytest tests/test_gpu_dropin.py -x -q -k "lzma" 2>&1 | grep -v amdgpu.ids | tail -2 ) > $out/check.log 2>&1
( timeout 200 python tests/fuzz_lzma_windows.py 6 4 2>&1 | grep -v amdgpu.ids | tail -2 ) >> $out/check.log 2>&1
cat $out/check.log

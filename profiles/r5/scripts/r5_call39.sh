#!/bin/bash
# Round 5, call 39: the window fuzz of the suite (8 streams, seed 21) with TOTAL_IN_MAX in its cases, on the device
set -u
root=$PWD; out=$root/gpurun_out/c39; mkdir -p $out
( timeout 200 python -m pytest tests/test_gpu_dropin.py -x -q -k "test_window_mode_differential_fuzz or code_length" 2>&1 | grep -v amdgpu.ids | tail -2 ) > $out/check.log 2>&1
cat $out/check.log

#!/bin/bash
# Round 5, call 41: the wrapper error cases (a gzip trailer cut inside ISIZE) on the device
set -u
root=$PWD; out=$root/gpurun_out/c41; mkdir -p $out
( timeout 100 python -m pytest tests/test_gpu_wrappers.py -x -q -k "error_parity or optional_header" 2>&1 | grep -v amdgpu.ids | tail -2 ) > $out/check.log 2>&1
cat $out/check.log

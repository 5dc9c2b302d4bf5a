#!/bin/bash
# Round 5, call 42: test_refusal_behind_a_full_buffer on the device
set -u
root=$PWD; out=$root/gpurun_out/c42; mkdir -p $out
( timeout 60 python -m pytest tests/test_gpu_dropin.py -x -q -k "refusal_behind or crafted" 2>&1 | grep -v amdgpu.ids | tail -2 ) > $out/check.log 2>&1
cat $out/check.log

#!/bin/bash
# Round 5, call 44: the early refusals of test_refusal_behind_a_full_buffer (the second decode behind make-believe history) on the device
set -u
root=$PWD; out=$root/gpurun_out/c44; mkdir -p $out
( timeout 9 python -m pytest tests/test_gpu_dropin.py -x -q -k "refusal_behind or crafted" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -2 ) > $out/check.log 2>&1
cat $out/check.log

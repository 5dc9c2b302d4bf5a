#!/bin/bash
# Round 5, call 40: the crafted corner streams (tests/test_gpu_dropin.py::test_crafted_deflate_corners) on the device
set -u
root=$PWD; out=$root/gpurun_out/c40; mkdir -p $out
( timeout 120 python -m pytest tests/test_gpu_dropin.py -x -q -k "crafted" 2>&1 | grep -v amdgpu.ids | tail -2 ) > $out/check.log 2>&1
cat $out/check.log

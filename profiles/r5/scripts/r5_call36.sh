#!/bin/bash
# Round 5, call 36: TOTAL_IN at a data error counts the bits of the code inflate() refuses (the step loop's verdicts) and the
# repeat symbol it has pulled ("invalid bit length repeat"): parity tests, the fuzz gate, the window fuzz, the probes
set -u
root=$PWD; out=$root/gpurun_out/c36; mkdir -p $out
{
timeout 400 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_dropin.py tests/test_gpu_wrappers.py -x -q -k "not xz and not lzma" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tests/fuzz_gpu.py 12000 7 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python tests/fuzz_gpu_windows.py 90 12 2>&1 | grep -v amdgpu.ids | tail -3
timeout 100 python tests/perf_probe.py 2>&1 | grep "rep 2"
timeout 100 python tests/perf_probe.py 512 200000 8192 2>&1 | grep "rep 2"
} > $out/check.log 2>&1
cat $out/check.log

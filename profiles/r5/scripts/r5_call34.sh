#!/bin/bash
# Round 5, call 34: the block header with a code-length code of no code at all (inflate() refuses it nlen + ndist bits later than
# K1 did: TOTAL_IN) -- only mz_block_code differs from the evidence build: parity tests, the fuzz gate, the two probes
set -u
root=$PWD; out=$root/gpurun_out/c34; mkdir -p $out
{
timeout 400 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_dropin.py -x -q -k "not xz and not lzma" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 200 python tests/fuzz_gpu.py 6000 5 2>&1 | grep -v amdgpu.ids | tail -1
timeout 100 python tests/perf_probe.py 2>&1 | grep "rep 2"
timeout 100 python tests/perf_probe.py 512 200000 8192 2>&1 | grep "rep 2"
} > $out/check.log 2>&1
cat $out/check.log

#!/bin/bash
# Round 5, call 8: the GPU suite from test_gpu_hash on (call 7 stopped there: its un-primed walk now needs MZHIP_AUTOPRIME=0)
set -u
root=$PWD; out=$root/gpurun_out/c8; mkdir -p $out
( timeout 1800 python -X faulthandler -m pytest tests/test_gpu_hash.py tests/test_gpu_inflate.py tests/test_gpu_lzma.py tests/test_gpu_lzma_enc.py tests/test_gpu_prime.py tests/test_gpu_prime_write.py tests/test_gpu_streams.py tests/test_gpu_wrappers.py tests/test_gpu_xz.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -25 ) > $out/gputest.log 2>&1
cat $out/gputest.log

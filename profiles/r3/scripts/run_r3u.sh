#!/bin/bash
set -u
o=gpurun_out/r3u; mkdir -p $o
( MZ_NEAR=16 MZ_MODES=2 timeout 100 python tests/perf_threads.py 2>&1 | grep -v amdgpu.ids ) > $o/threads.log 2>&1
( timeout 200 python -m pytest tests/test_gpu_prime.py -x -q 2>&1 | tail -3 ) > $o/tests.log 2>&1
grep "^mode" $o/threads.log; cat $o/tests.log

#!/bin/bash
set -u
o=gpurun_out/r3t; mkdir -p $o
( MZ_NEAR=16 MZ_MODES=2 MZDROP_TRACE=1 MZHIP_PRIME_TRACE=1 timeout 100 python tests/perf_threads.py 2>&1 | grep -v amdgpu.ids ) > $o/threads.log 2>&1
grep "^mode" $o/threads.log

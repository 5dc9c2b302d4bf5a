#!/bin/bash
# Round 5, call 18: the prime pipeline's timeline under the readers (MZHIP_PRIME_TRACE=2: when every chunk becomes servable)
set -u
root=$PWD; out=$root/gpurun_out/c18; mkdir -p $out
( MZ_NEAR=16 MZ_MODES=2 MZHIP_PRIME_TRACE=2 MZDROP_TRACE=1 timeout 600 python tests/perf_threads.py 2>&1 | grep -v amdgpu.ids ) > $out/threads_trace.log 2>&1
tail -60 $out/threads_trace.log

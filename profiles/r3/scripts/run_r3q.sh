#!/bin/bash
# round-3 experiment q: progressive prime + chunked record layout (one gpurun call)
set -u
o=gpurun_out/r3q; mkdir -p $o
( timeout 400 python -m pytest tests/test_gpu_prime.py tests/test_gpu_inflate.py tests/test_gpu_dropin.py -x -q -s 2>&1 | tail -25 ) > $o/tests.log 2>&1
( echo "== chunked (default)"; timeout 60 python tests/perf_probe.py 2>&1 | grep -v 'amdgpu.ids'; timeout 60 python tests/perf_probe.py 512 200000 8192 2>&1 | grep -v 'amdgpu.ids' ) > $o/ab.log 2>&1
( timeout 200 bash profiles/ab_k1.sh run ) >> $o/ab.log 2>&1
( MZDROP_TRACE=1 MZHIP_PRIME_TRACE=1 timeout 200 python tests/perf_threads.py 2>&1 | grep -v 'amdgpu.ids' | tail -60 ) > $o/threads.log 2>&1
( timeout 400 python bench.py 2> $o/bench.err ) > $o/bench.log
tail -5 $o/tests.log; cat $o/ab.log | grep "rep 2\|passed\|failed\|=="; grep "^mode" $o/threads.log; cat $o/bench.log

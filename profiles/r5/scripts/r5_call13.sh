#!/bin/bash
# round 5, GPU call 13: method 95 READ in window mode on the device (unit test, differential fuzz, 3 GiB entry with the RSS
# bound), the ABI test, and the rank tests of call 12 that failed on a missing key
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c13
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dropin.py -q -x -m gpu -s -k "xz_window or lzma_window" > gpurun_out/c13/xz_tests.log 2>&1
echo "xz tests rc=$?"; tail -4 gpurun_out/c13/xz_tests.log
timeout 1700 python -m pytest tests/test_gpu_streams.py -q -x -m gpu -s -k "xz_entry_larger" > gpurun_out/c13/xz_big.log 2>&1
echo "xz big rc=$?"; tail -4 gpurun_out/c13/xz_big.log
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_xz.py -q -x -m gpu > gpurun_out/c13/ranks.log 2>&1
echo "ranks rc=$?"; tail -3 gpurun_out/c13/ranks.log

#!/bin/bash
# Round 5, call 14: near copies with rank-interval dependency masks instead of the pending-bit map (same rounds, fewer
# instructions); the CRC super-tile advance as four global table lookups instead of 32 select-xor steps
set -u
root=$PWD; out=$root/gpurun_out/c14; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in near nearcrc; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in base near nearcrc base near; do echo "== $t 64K"; probe $t; done
for t in base near nearcrc base; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
cat $out/probe.log

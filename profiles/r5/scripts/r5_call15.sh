#!/bin/bash
# Round 5, call 15: the latency-bound copy phases of K1 once more, now that the wave's control is scalar: near copies with every
# load before the first store (one LDS round trip instead of up to seven), far copies with two batches in flight, the CRC advance
# as four global lookups
set -u
root=$PWD; out=$root/gpurun_out/c15; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in nearld far2ld nearcrc; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in near nearld far2 far2ld nearcrc near; do echo "== $t 64K"; probe $t; done
for t in near nearld far2 far2ld nearcrc near; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
cat $out/probe.log

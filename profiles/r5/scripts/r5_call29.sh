#!/bin/bash
# Round 5, call 29: K1's emit round writes its back-references as TWO lists (far entries / near entries, classified by the lanes
# while the records become bytes) instead of one that every piece went through the far loop for ("split"), and the two copies
# (output buffer -> staging, staging -> staging) in three cases by size with clamped offsets instead of 4 x 8 + 4 + 2 + 1 bytes
# under an exec bracket each ("diet" = split + that)
set -u
root=$PWD; out=$root/gpurun_out/c29; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in diet; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_streams.py -x -q -k "not bounded_memory" 2>&1 | tail -2; done
for t in base split diet base split diet; do echo "== $t 64K"; probe $t; done
for t in base split diet base diet; do echo "== $t 8K"; probe $t 512 200000 8192; done
echo "== diet fuzz"; MZHIP_LIB=$B/_build_ab_diet/libmzhip.so timeout 300 python tests/fuzz_gpu.py 3000 11 2>&1 | tail -2
} > $out/probe.log 2>&1
cat $out/probe.log

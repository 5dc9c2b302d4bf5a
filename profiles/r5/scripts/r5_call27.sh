#!/bin/bash
# Round 5, call 27: a fifth wave per SIMD for K1.  The walk's per-lane stream rings shrink from 16 to 8 dwords (topped up 8 bytes at
# a time, checked before every step: +8 % vector instructions per trip of pass 1), which takes a wave's LDS from 9 984 to 7 936
# bytes = five workgroups of four waves per CU; the kernel's registers fit 96 (walk 72, emit 88; 21 spilled in the entry's own code).
#   base   HEAD (16-dword rings, 4 waves per SIMD)      r8w4  8-dword rings at 4 waves per SIMD (what the rings alone cost)
#   r8w5   8-dword rings, 5 waves per SIMD              r8w5p the same with 256 bytes less pool (if five do not fit at exactly 160 KiB)
set -u
root=$PWD; out=$root/gpurun_out/c27; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in r8w5; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_streams.py -x -q -k "not bounded_memory" 2>&1 | tail -2; done
for t in base r8w4 r8w5 base r8w4 r8w5; do echo "== $t 64K"; probe $t; done
for t in base r8w5 r8w4 base r8w5; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
if ! grep -q "geometry: 1280 resident" $out/probe.log; then # five workgroups per CU did not fit at exactly 160 KiB: 256 bytes less pool
  { for t in r8w5p r8w5p; do echo "== $t 64K"; probe $t; done; echo "== r8w5p 8K"; probe r8w5p 512 200000 8192; } >> $out/probe.log 2>&1
fi
cat $out/probe.log

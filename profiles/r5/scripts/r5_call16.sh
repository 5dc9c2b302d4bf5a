#!/bin/bash
# Round 5, call 16: CRC of K1 with fewer instructions -- the advance as global lookups (nearcrc, +2.7 % in call 15), the dword step as
# four independent global lookups in the slicing tables (crcg4), two far batches in flight on top of the advance (crcfar)
set -u
root=$PWD; out=$root/gpurun_out/c16; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in crcg4 crcfar; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in nearcrc crcg4 crcfar nearcrc crcg4 crcfar; do echo "== $t 64K"; probe $t; done
for t in nearcrc crcg4 crcfar nearcrc; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
cat $out/probe.log

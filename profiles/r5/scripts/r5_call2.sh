#!/bin/bash
# Round 5, call 2: (a) the wave-index uniformity fix alone (uni: round-4 header, MZ_HDR_V2=0) and (b) with the round-5 block
# header on top (hdr2), against the build of call 1 (base): parity (tests/test_gpu_inflate.py) + the probes + per-section cycles
set -u
root=$PWD; out=$root/gpurun_out/c2; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in uni hdr2; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in base uni hdr2 hdr2prof; do echo "== $t 64K"; probe $t; done
for t in base uni hdr2 hdr2prof; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
cat $out/probe.log

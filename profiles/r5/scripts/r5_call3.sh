#!/bin/bash
# Round 5, call 3: hdr3 = uniformity fix + round-5 CLC build and tables (inflate_tables.inc) behind the round-4 code-length front end
# + emit rounds cut at whole lanes; against uni (uniformity fix only) and base
set -u
root=$PWD; out=$root/gpurun_out/c3; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in hdr3; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in base uni hdr3 hdr3prof; do echo "== $t 64K"; probe $t; done
for t in base uni hdr3 hdr3prof; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
cat $out/probe.log

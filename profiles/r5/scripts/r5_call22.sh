#!/bin/bash
# Round 5, call 21: pass 1 / pass 2 of K1 with fewer instructions per trip (one loop exit at the top instead of a second in the body:
# ~20 register copies a trip; the refill's prefetch reloaded by every lane: no merge copies; one exec bracket for the leading literal;
# leaner per-step bookkeeping): 604 -> 496 vector instructions per trip of pass 1, 895 -> 805 of pass 2
set -u
root=$PWD; out=$root/gpurun_out/c22; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in walk3; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in base walk3 walk2 base walk3; do echo "== $t 64K"; probe $t; done
for t in base walk3 walk2 base walk3; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
cat $out/probe.log

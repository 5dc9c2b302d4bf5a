#!/bin/bash
# Round 5, call 23: the block's code (code lengths -> decode tables) as a function of its own (mz_block_code): the header's state no
# longer competes with the entry's for scalar registers -- the code-length window loop 956 -> 193 instructions, K1's scratch 220 -> 40 B
set -u
root=$PWD; out=$root/gpurun_out/c23; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in hdrfn; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_streams.py -x -q -k "not bounded_memory" 2>&1 | tail -2; done
for t in base hdrfn base hdrfn; do echo "== $t 64K"; probe $t; done
for t in base hdrfn base hdrfn; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
cat $out/probe.log

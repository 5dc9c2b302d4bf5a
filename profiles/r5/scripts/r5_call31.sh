#!/bin/bash
# Round 5, call 31: K3 without the end-of-input flag (the input position one past the end says it: one value less for every
# normalisation site of the unrolled trees to merge) -- "eof" -- and with range <<= 8 as a tied one-instruction update -- "eoft"
set -u
root=$PWD; out=$root/gpurun_out/c31; mkdir -p $out
B=$root/minizip-ng_amd
{
echo "== eoft parity"; MZHIP_LIB=$B/_build_ab_eoft/libmzhip.so timeout 500 python -m pytest tests/test_gpu_lzma.py tests/test_gpu_xz.py -x -q 2>&1 | tail -2
for t in base eof eoft base eof eoft; do echo "== $t lzma 4096 x 1 MiB"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 200 python tests/perf_codecs.py lzma 4096 2>&1 | grep -v amdgpu.ids | tail -2; done
} > $out/probe.log 2>&1
cat $out/probe.log

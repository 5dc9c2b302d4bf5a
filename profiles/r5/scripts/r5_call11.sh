#!/bin/bash
# Round 5, call 11: branches in the header path -- the code-length front end's walk unrolled without a branch (16 / 24 links), the
# table build's ballot loop unrolled without its skip branch; probes on 8 KiB and 64 KiB entries
set -u
root=$PWD; out=$root/gpurun_out/c11; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in clu24 tabbl; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in head clu16 clu24 tabbl both head; do echo "== $t 8K"; probe $t 512 200000 8192; done
for t in head clu24 tabbl both; do echo "== $t 64K"; probe $t; done
} > $out/probe.log 2>&1
cat $out/probe.log

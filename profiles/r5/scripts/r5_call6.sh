#!/bin/bash
# Round 5, call 6: K1's fused CRC with 2 / 4 slicing tables (pool 1264 -> 1008 / 496 bytes per wave) against the pool shrink alone;
# K3 / K4 on HEAD against the round-4 build on one box (the wave-index fix touched them)
set -u
root=$PWD; out=$root/gpurun_out/c6; mkdir -p $out
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in sl2 sl4; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in hdr3 pool496 sl2 sl4; do echo "== $t 64K"; probe $t; done
for t in hdr3 pool496 sl2 sl4; do echo "== $t 8K"; probe $t 512 200000 8192; done
for t in base hdr3; do for c in 4 5; do echo "== $t config $c"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
} > $out/probe.log 2>&1
cat $out/probe.log

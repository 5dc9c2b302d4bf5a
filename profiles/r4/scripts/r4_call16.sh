#!/bin/bash
# Round 4, call 16: K3's bit trees with the children fetched as one word before the node is decided (MZ_LZMA_PAIRS) against
# the build without it, on config 4 and the word-salad probe; K3 / .xz tests on the device; K1's sections on 8 KiB entries
set -u
mkdir -p gpurun_out/c16
python -c "import torch" 2>/dev/null
for tag in k3_pairs0 default; do
  lib=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ $tag = default ] && lib=$PWD/minizip-ng_amd/_build/libmzhip.so
  echo "== $tag"
  MZHIP_LIB=$lib timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['crc32_match_rate'])"
  MZHIP_LIB=$lib timeout 120 python tests/perf_codecs.py lzma 4096 2>&1 | grep "LZMA decode"
  MZHIP_LIB=$lib timeout 120 python tests/perf_codecs.py xz 2>&1 | grep -i "xz"
done > gpurun_out/c16/ab_k3_pairs.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_lzma.py tests/test_gpu_xz.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3 ) > gpurun_out/c16/test_lzma.log 2>&1
( echo "== K1 sections, 200000 x 8 KiB"; MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_prof/libmzhip.so timeout 100 python tests/perf_probe.py 512 200000 8192 2>&1 | grep -v '^rep [01]\|amdgpu.ids'
  echo "== K1 sections, 20000 x 64 KiB"; MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_prof/libmzhip.so timeout 100 python tests/perf_probe.py 2>&1 | grep -v '^rep [01]\|amdgpu.ids' ) > gpurun_out/c16/k1_sections.log 2>&1
cat gpurun_out/c16/*.log

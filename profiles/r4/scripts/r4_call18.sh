#!/bin/bash
# Round 4, call 18: K3's slot kernel with 1, 2, 4 of every four waves of a SIMD on the scalar-port form (the PMC pass of
# call 17 says the vector port is 89 % busy and the scalar one 32 %)
set -u
mkdir -p gpurun_out/c18
python -c "import torch" 2>/dev/null
for tag in default k3_sport1 k3_sport2 k3_sport4; do
  lib=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ $tag = default ] && lib=$PWD/minizip-ng_amd/_build/libmzhip.so
  echo "== $tag"
  MZHIP_LIB=$lib timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['crc32_match_rate'])"
  MZHIP_LIB=$lib timeout 120 python tests/perf_codecs.py lzma 4096 2>&1 | grep "LZMA decode"
done > gpurun_out/c18/ab_k3_ports.log 2>&1
cat gpurun_out/c18/*.log

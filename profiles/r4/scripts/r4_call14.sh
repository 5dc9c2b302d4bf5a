#!/bin/bash
# Round 4, call 14: K3's decisions asked through a ballot (MZ_LZMA_UBR: 0 = round 3's exec-masked diamonds, 1 = uniform
# branches, 2 = the probability on the scalar unit as well) on config 4 and on the word-salad probe
set -u
mkdir -p gpurun_out/c14
python -c "import torch" 2>/dev/null
for tag in k3_ubr0 default k3_ubr2; do
  lib=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ $tag = default ] && lib=$PWD/minizip-ng_amd/_build/libmzhip.so
  echo "== $tag"
  MZHIP_LIB=$lib timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['crc32_match_rate'])"
  MZHIP_LIB=$lib timeout 120 python tests/perf_codecs.py lzma 4096 2>&1 | grep "LZMA decode"
done > gpurun_out/c14/ab_k3_ubr.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_lzma.py tests/test_gpu_xz.py -x -q 2>&1 | tail -3 ) > gpurun_out/c14/test_lzma.log 2>&1
cat gpurun_out/c14/*.log

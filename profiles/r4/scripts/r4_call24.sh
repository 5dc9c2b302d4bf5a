#!/bin/bash
# Round 4, call 24: K3's decisions with `code < bound` and `code - bound` from one subtraction (MZ_LZMA_BORROW) against the
# build without it (config 4, word-salad probe); then the whole GPU suite on HEAD
set -u
mkdir -p gpurun_out/c24
python -c "import torch" 2>/dev/null
for tag in k3_borrow0 default; do
  lib=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ $tag = default ] && lib=$PWD/minizip-ng_amd/_build/libmzhip.so
  echo "== $tag"
  MZHIP_LIB=$lib timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['crc32_match_rate'])"
  MZHIP_LIB=$lib timeout 120 python tests/perf_codecs.py lzma 4096 2>&1 | grep "LZMA decode"
done > gpurun_out/c24/ab_k3_borrow.log 2>&1
cat gpurun_out/c24/ab_k3_borrow.log
( timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -8 ) > gpurun_out/c24/gputest.log 2>&1
cat gpurun_out/c24/gputest.log

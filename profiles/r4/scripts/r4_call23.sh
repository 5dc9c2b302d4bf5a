#!/bin/bash
# Round 4, call 23: K6's range coder with the probabilities of a symbol's path read up front (MZ_LZE_PRELOAD) against the
# build without it on the encode probe; the encoder's tests on the device
set -u
mkdir -p gpurun_out/c23
python -c "import torch" 2>/dev/null
for tag in k6_pre0 default; do
  lib=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ $tag = default ] && lib=$PWD/minizip-ng_amd/_build/libmzhip.so
  echo "== $tag"
  MZHIP_LIB=$lib timeout 300 python tests/perf_codecs.py lzmaenc 2>&1 | grep "LZMA encode"
done > gpurun_out/c23/ab_k6_preload.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_lzma_enc.py -x -q 2>&1 | tail -3 ) > gpurun_out/c23/test_lzma_enc.log 2>&1
cat gpurun_out/c23/*.log

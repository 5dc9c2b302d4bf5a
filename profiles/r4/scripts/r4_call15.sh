#!/bin/bash
# Round 4, call 15: K6 with the chain pass on the device -- its tests (ratio bar on config-4 entries, host = batch bytes,
# segments = one piece), the encode probe against the build before it (_build_ab_k3_ubr0 = round 3's K6)
set -u
mkdir -p gpurun_out/c15
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests/test_gpu_lzma_enc.py tests/test_gpu_xz.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/c15/test_lzma_enc.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_prime_write.py -x -q -k "lzma or write or segment" 2>&1 | tail -5 ) > gpurun_out/c15/test_dropin.log 2>&1
for tag in k3_ubr0 default; do
  lib=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ $tag = default ] && lib=$PWD/minizip-ng_amd/_build/libmzhip.so
  echo "== $tag"
  MZHIP_LIB=$lib timeout 300 python tests/perf_codecs.py lzmaenc 2>&1 | grep "LZMA encode"
done > gpurun_out/c15/perf_lzmaenc.log 2>&1
cat gpurun_out/c15/*.log

#!/bin/bash
# Round 4, call 25: K6's classes on 2304 x 1 MiB (preset 1 = one candidate, one link; 4 / 6 / 9 = four candidates and 4 / 8 / 16
# links of the chain), then the encoder's tests
set -u
mkdir -p gpurun_out/c25
python -c "import torch" 2>/dev/null
for p in 1 4 6 9; do
  LZMA_PRESET=$p timeout 120 python tests/perf_codecs.py lzmaenc 2>&1 | grep "LZMA encode"
done > gpurun_out/c25/k6_presets.log 2>&1
cat gpurun_out/c25/k6_presets.log
( timeout 300 python -m pytest tests/test_gpu_lzma_enc.py -x -q 2>&1 | tail -3 ) > gpurun_out/c25/test_lzma_enc.log 2>&1
cat gpurun_out/c25/test_lzma_enc.log

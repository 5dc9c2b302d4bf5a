#!/bin/bash
# Round 4, call 19: balanced rounds (grid_for) on one rank's share of the 8-GPU run and on the whole table, against the
# build before (_build_ab_k3_pairs0 launches every resident wave); then the whole GPU suite on HEAD
set -u
mkdir -p gpurun_out/c19
python -c "import torch" 2>/dev/null
for tag in k3_pairs0 default; do
  lib=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so
  [ $tag = default ] && lib=$PWD/minizip-ng_amd/_build/libmzhip.so
  for e in 12500 25000 50000 100000; do
    echo "== $tag, $e entries"
    MZHIP_LIB=$lib timeout 300 python bench.py --config 2 --entries $e --steps 10 --warmup 3 --no-legs --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['crc32_match_rate'], d['config']['launch'])"
  done
done > gpurun_out/c19/ab_balanced_rounds.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/c19/gputest.log 2>&1
cat gpurun_out/c19/*.log

#!/bin/bash
# the last seconds of the round's GPU budget: first timing of the two K1 knobs on the 20 000 x 64 KiB probe
o=gpurun_out/r3w; mkdir -p $o
for v in base sel pre2; do
  echo "== $v" >> $o/ab.log
  MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_$v/libmzhip.so timeout 14 python tests/perf_probe.py 2>&1 | grep "rep 2\|rep 1" >> $o/ab.log
done
cat $o/ab.log

#!/bin/bash
# Round 5, call 37: the bench line of the final binary (config 2 only: no legs, no other configs, no CPU baseline)
set -u
root=$PWD; out=$root/gpurun_out/c37; mkdir -p $out
( timeout 300 python bench.py --no-legs --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 ) > $out/bench_cfg2.log
cut -c1-300 $out/bench_cfg2.log

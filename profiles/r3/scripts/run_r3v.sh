#!/bin/bash
# the build that ships, after the evidence run: the GPU suite once more, the DEFLATE classes (four-chain cost parse)
set -u
o=gpurun_out/final2; mkdir -p $o
( timeout 300 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -14 ) > $o/gputest.log 2>&1
( timeout 60 python tests/perf_codecs.py deflate_levels 2>&1 | grep -v amdgpu.ids ) > $o/k4_levels.log 2>&1
( MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_k4prof/libmzhip.so timeout 60 python tests/perf_codecs.py deflate_levels 2>&1 | grep -v amdgpu.ids ) > $o/k4_levels_sections.log 2>&1
cat $o/gputest.log; grep "DEFLATE encode" $o/k4_levels.log

#!/bin/bash
set -u
o=gpurun_out/r3r; mkdir -p $o
nproc > $o/env.log; cat /sys/fs/cgroup/cpu.max >> $o/env.log; lscpu | grep -i "numa\|socket\|thread\|model name" >> $o/env.log; rocm-smi --showtopo 2>/dev/null | tail -12 >> $o/env.log
( timeout 200 python tests/perf_h2d.py 2>&1 | grep -v 'amdgpu.ids' ) > $o/h2d.log 2>&1
( MZ_MODES=2 timeout 100 python tests/perf_threads.py 2>&1 | grep "^mode\|pinned" ) > $o/threads.log 2>&1
for pin in 0-15 0-31 64-79; do ( MZ_PIN=$pin MZ_MODES=2 timeout 100 python tests/perf_threads.py 2>&1 | grep "^mode\|pinned" ) >> $o/threads.log 2>&1; done
cat $o/env.log $o/h2d.log $o/threads.log

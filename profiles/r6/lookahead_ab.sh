#!/bin/bash
# gpurun -- 'bash profiles/r6/lookahead_ab.sh <name>': window mode with and without the one-window look-ahead (MZHIP_STREAM_LOOKAHEAD=0),
# a 1 GiB entry of each kind, then the device's window tests and 120 fuzz streams with the look-ahead on.
out=gpurun_out/${1:-la}; mkdir -p $out
for la in 1 0; do for kind in text mixed sparse; do echo "== look-ahead $la, $kind"
  MZHIP_STREAM_LOOKAHEAD=$la MZHIP_STREAM_STATS=1 timeout 300 python tests/perf_large_entry.py $kind 1 2>&1 | grep -v amdgpu.ids
done; done > $out/large_entry_full.log 2>&1
grep -v "serial windows [12][0-9] " $out/large_entry_full.log | grep -v "^many-wave decode OFF" > $out/large_entry.log
timeout 600 python -m pytest tests/test_gpu_streams.py -x -q 2>&1 | tail -2 > $out/tests.log
timeout 600 python tests/fuzz_gpu_windows.py 120 7 2>&1 | tail -3 >> $out/tests.log
cat $out/tests.log $out/large_entry.log

#!/bin/bash
# Round 4, call 8: the many-wave decode of one large entry on the device for the first time; the WRITE_SIZE passes of
# configs 2 and 5 that hung in the evidence run; the calibration with kernels of known traffic at 1 / 4 / 16 B per lane.
set -u
mkdir -p gpurun_out/c8
python -c "import torch" 2>/dev/null
( timeout 300 python -m pytest tests/test_gpu_streams.py -x -q -s -k "one_window" 2>&1 | grep -v amdgpu.ids | tail -15 ) > gpurun_out/c8/one_window.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -s -k "many_waves or window_mode" 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/c8/dropin.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_streams.py -x -q -s -k "larger_than_any_window" 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/c8/large.log 2>&1
cat gpurun_out/c8/one_window.log gpurun_out/c8/dropin.log gpurun_out/c8/large.log
export TMPDIR=/tmp
root=$PWD
( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do lc=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/c8/cal_$lc -o pmc --output-format csv -- python $root/profiles/pmc_calibrate.py > $root/gpurun_out/c8/cal_$lc.log 2>&1
    find $root/gpurun_out/c8/cal_$lc -name '*counter_collection.csv' -exec cp {} $root/gpurun_out/c8/cal_$lc.csv \;
    rm -rf $root/gpurun_out/c8/cal_$lc
  done )
MZ_COLLECT_TIMEOUT=150 bash profiles/collect.sh c8_w2 WRITE_SIZE > gpurun_out/c8/collect_w2.log 2>&1
MZ_COLLECT_CONFIG=5 MZ_COLLECT_KERNEL=k_deflate_batch MZ_COLLECT_TIMEOUT=150 bash profiles/collect.sh c8_w5 WRITE_SIZE > gpurun_out/c8/collect_w5.log 2>&1
ls -la gpurun_out/c8 gpurun_out/c8_w2 gpurun_out/c8_w5 | head -40

#!/bin/bash
# Round 5, call 5: the whole GPU suite on HEAD (the wave-index fix touched five kernels), the default bench line, the clock under K1
set -u
root=$PWD; out=$root/gpurun_out/c5; mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 ) > $out/gputest.log 2>&1
cat $out/gputest.log
( timeout 600 python bench.py 2>$out/bench.err | tail -1 ) > $out/bench.log
cat $out/bench.log
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $out/clk -o pmc --output-format csv -- python $root/tests/perf_probe.py > $out/clk.log 2>&1
find $out/clk -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_inflate_batch" "$1" > "$2"' _ {} $out/clk.csv \;
rm -rf $out/clk
cat $out/clk.csv | cut -d, -f16-19

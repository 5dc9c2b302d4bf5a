#!/bin/bash
# Round 5, call 28: did call 27's five workgroups per CU really run side by side?  Average resident waves of K1 (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES)
# for base, r8w5 (LDS exactly 160 KiB per CU) and r8w5p (256 bytes less per wave); r8w5p on the probe
set -u
root=$PWD; out=$root/gpurun_out/c28; mkdir -p $out
B=$root/minizip-ng_amd
export TMPDIR=/tmp
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in r8w5p base r8w5p; do echo "== $t 64K"; probe $t; done
} > $out/probe.log 2>&1
cd /tmp
for t in base r8w5 r8w5p; do
  MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/occ_$t -o pmc --output-format csv -- python $root/tests/perf_probe.py > $out/occ_$t.log 2>&1
  find $out/occ_$t -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_inflate_batch" "$1" > "$2"' _ {} $out/occ_$t.csv \;
  rm -rf $out/occ_$t
done
cat $out/probe.log; ls $out

#!/bin/bash
# Round 5, call 26: K1's instruction mix and busy cycles on HEAD (20 000 x 64 KiB probe, 200 000 x 8 KiB), one --pmc pass per group
set -u
root=$PWD; out=$root/gpurun_out/c26; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
sq() { # tag, cmd...
  tag=$1; shift; i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $out/sq_${tag}_$i -o pmc --output-format csv -- "$@" > $out/sq_${tag}_$i.log 2>&1
    find $out/sq_${tag}_$i -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_inflate_batch" "$1" > "$2"' _ {} $out/sq_${tag}_$i.csv \;
    rm -rf $out/sq_${tag}_$i
  done
}
sq 64k python $root/tests/perf_probe.py
sq 8k python $root/tests/perf_probe.py 512 200000 8192
ls $out

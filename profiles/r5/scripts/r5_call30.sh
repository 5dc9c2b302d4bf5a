#!/bin/bash
# Round 5, call 30: is K1 short of LDS cycles?  LDS counters of HEAD on the 64 KiB probe (bank conflicts, index-active cycles)
set -u
root=$PWD; out=$root/gpurun_out/c30; mkdir -p $out
B=$root/minizip-ng_amd
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_BUSY_CU[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_INSTS_SMEM\|SQ_WAIT_INST[A-Z_]*\|TA_BUSY[a-z_]*\|TCP_[A-Z_]*STALL[A-Z_]*" | sort -u > $out/avail.txt
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES" \
           "SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  MZHIP_LIB=$B/_build_ab_base/libmzhip.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $out/lds_$i -o pmc --output-format csv -- python $root/tests/perf_probe.py > $out/lds_$i.log 2>&1
  find $out/lds_$i -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_inflate_batch" "$1" > "$2"' _ {} $out/lds_$i.csv \;
  rm -rf $out/lds_$i
done
cat $out/avail.txt | tr '\n' ' '; tail -3 $out/lds_1.log; ls $out

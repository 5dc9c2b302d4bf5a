#!/bin/bash
# Counter passes on the short probe (tests/perf_probe.py: 20 000 x 64 KiB, 3 launches): HBM traffic and the SQ groups of
# profiles/collect.sh, one rocprofv3 --pmc pass per group (never combined with other trace domains).
#   gpurun -- 'bash profiles/pmc_probe.sh <tag>'   -> gpurun_out/<tag>/pmc_*.csv
set -u
tag=${1:-pmc}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  timeout -k 10 ${MZ_COLLECT_TIMEOUT:-200} rocprofv3 --kernel-trace --pmc $grp -d "$out/p$i" -o pmc --output-format csv -- python $root/tests/perf_probe.py ${MZ_PROBE_ARGS:-} > "$out/p$i.log" 2>&1
  find "$out/p$i" -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_inflate_batch" "$1" > "$2"' _ {} "$out/pmc_$i.csv" \;
  rm -rf "$out/p$i"
done
python3 - "$out" <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
for f in sorted(glob.glob(out+'/pmc_*.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print("%-28s launches %d  last %.4g" % (k,len(v),v[-1]))
PY

#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag>      -> gpurun_out/<tag>/{bench.log,kernel_stats.csv,pmc_fetch_size.csv,pmc_write_size.csv}
# Three separate runs of the same bench command: plain, rocprofv3 --kernel-trace --stats, and one --pmc pass per
# counter (never combined with other trace domains).  Copy what should be judged into profiles/<round>/.
set -u
tag=${1:-run}
what=${2:-all}   # all | bench | stats | FETCH_SIZE | WRITE_SIZE | SQ (instruction mix and wait cycles) | REQ (L2 memory-side requests by size); the last two are not part of `all`
T=${MZ_COLLECT_TIMEOUT:-300}  # a pass that outlives this is killed (rocprofv3 has hung here once)
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cfgflag=${MZ_COLLECT_CONFIG:+--config $MZ_COLLECT_CONFIG}   # default: the headline config
kern=${MZ_COLLECT_KERNEL:-k_inflate_batch}                  # the kernel whose counter rows are kept
cmd="python $root/bench.py $cfgflag --steps 3 --warmup 1"
if [ $what = all ] || [ $what = bench ]; then ( cd "$root" && timeout $T $cmd > "$out/bench.log" 2> "$out/bench.err" ); tail -1 "$out/bench.log"; fi
# the profiled passes launch the dominant kernel on the bench workload only (no legs, no CPU baseline), so that the
# per-kernel average of --stats is the average of identical launches: 1 warm-up + 3 timed
cmd="$cmd --no-legs --no-cpu-baseline --no-other-configs"
cd /tmp
if [ $what = all ] || [ $what = stats ]; then
  timeout -k 10 $T rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats --output-format csv -- $cmd > "$out/stats.log" 2>&1
  find "$out/stats" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
fi
for c in FETCH_SIZE WRITE_SIZE; do
  if [ $what != all ] && [ $what != $c ]; then continue; fi
  lc=$(echo $c | tr 'A-Z' 'a-z')
  timeout -k 10 $T rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_$lc" -o pmc --output-format csv -- $cmd > "$out/pmc_$lc.log" 2>&1
  find "$out/pmc_$lc" -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|$3" "$1" > "$2"' _ {} "$out/pmc_$lc.csv" "$kern" \;
done
if [ $what = REQ ]; then
  # the memory-side requests of the L2 by size: what FETCH_SIZE / WRITE_SIZE are derived from, for a kernel whose accesses are
  # not the wide streaming ones the guide's factor was found on (bytes = 32 x n32 + 64 x n64 + 128 x n128)
  i=0
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    reqcmd=${MZ_REQ_CMD:-$cmd}
    timeout -k 10 $T rocprofv3 --kernel-trace --pmc $grp -d "$out/pmc_req$i" -o pmc --output-format csv -- $reqcmd > "$out/pmc_req$i.log" 2>&1
    find "$out/pmc_req$i" -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|$3" "$1" > "$2"' _ {} "$out/pmc_req$i.csv" "${MZ_REQ_KERNEL:-$kern}" \;
  done
fi
if [ $what = SQ ]; then
  # instruction mix and stall picture of the dominant kernel; a pass per counter group (never with other trace domains);
  # MZ_SQ_CMD overrides the workload (default: the short probe, one launch is enough for counters)
  sqcmd=${MZ_SQ_CMD:-python $root/tests/perf_probe.py}
  sqk=${MZ_SQ_KERNEL:-k_inflate_batch}   # the kernel whose rows are kept
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_DATA_FIFO_FULL"; do
    i=$((i+1))
    timeout -k 10 $T rocprofv3 --kernel-trace --pmc $grp -d "$out/pmc_sq$i" -o pmc --output-format csv -- $sqcmd > "$out/pmc_sq$i.log" 2>&1
    find "$out/pmc_sq$i" -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|$3" "$1" > "$2"' _ {} "$out/pmc_sq$i.csv" "$sqk" \;
  done
fi
ls -la "$out"
head -3 "$out/kernel_stats.csv"

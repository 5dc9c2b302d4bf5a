#!/bin/bash
# Round 5, call 1: where HEAD stands on this box, and where K1's L2-miss bytes come from.
#   a. base build on the probe (20 000 x 64 KiB, 200 000 x 8 KiB), per-section cycles of the PROF build on both
#   b. existing knobs once more as a calibration of "latency or instruction count": NBATCH=1, FSLOTS=2, SLOTS=2
#   c. the L2's memory-side requests by size (TCC_EA0_RDREQ_{sum,32B,64B,128B}, TCC_EA0_WRREQ_{sum,64B}, + TCC_HIT/MISS) of
#      k_inflate_batch on 40 000 x 64 KiB (2048 unique streams) for HEAD and for HEAD minus far loads / CRC / record
#      stores / all match copies / the store; the same on bench.py's config 2 for HEAD
set -u
root=$PWD; out=$root/gpurun_out/c1; mkdir -p $out
export TMPDIR=/tmp
B=$PWD/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in base prof nbatch fslots2 slots2; do echo "== $t 64K"; probe $t; done
for t in base prof; do echo "== $t 8K"; probe $t 512 200000 8192; done
} > $out/probe.log 2>&1
cat $out/probe.log
cd /tmp
req() { # tag, cmd...
  tag=$1; shift
  i=0
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $out/req_${tag}_$i -o pmc --output-format csv -- "$@" > $out/req_${tag}_$i.log 2>&1
    find $out/req_${tag}_$i -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|k_inflate_batch" "$1" > "$2"' _ {} $out/req_${tag}_$i.csv \;
    rm -rf $out/req_${tag}_$i
  done
}
for t in base abl_far abl_crc abl_rec abl_copy abl_store; do
  MZHIP_LIB=$B/_build_ab_$t/libmzhip.so req $t python $root/tests/perf_probe.py 2048 40000
done
MZHIP_LIB=$B/_build_ab_base/libmzhip.so req cfg2 python $root/bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-other-configs
ls -la $out | head -40

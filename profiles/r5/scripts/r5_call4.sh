#!/bin/bash
# Round 5, call 4: where K1's step records live (MZ_REC_LAYOUT 0 = [quad][lane], 1 = a row per lane, 2 / 3 = [2 / 4 quads][lane]) and
# smaller windows (MZ_CHASE_SMAX 1024): time on the probes, and the L2's memory-side requests of the two extremes
set -u
root=$PWD; out=$root/gpurun_out/c4; mkdir -p $out
export TMPDIR=/tmp
B=$root/minizip-ng_amd
probe() { MZHIP_LIB=$B/_build_ab_$1/libmzhip.so timeout 120 python tests/perf_probe.py ${@:2} 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; }
{
for t in rl1 smax1k; do echo "== $t parity"; MZHIP_LIB=$B/_build_ab_$t/libmzhip.so timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2; done
for t in hdr3 rl1 rl2 rl3 smax1k smax1k_rl1; do echo "== $t 64K"; probe $t; done
for t in hdr3 rl1 rl2 rl3 smax1k smax1k_rl1; do echo "== $t 8K"; probe $t 512 200000 8192; done
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
for t in hdr3 rl1 rl3 smax1k; do
  MZHIP_LIB=$B/_build_ab_$t/libmzhip.so req $t python $root/tests/perf_probe.py 2048 40000
done
ls $out | head -40

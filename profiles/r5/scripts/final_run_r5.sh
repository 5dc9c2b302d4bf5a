#!/bin/bash
# Everything round 5's evidence consists of, in one gpurun call (from the repo root):
#   gpurun --timeout 3300 -- 'bash profiles/final_run_r5.sh'
# -> gpurun_out/final5/ ; profiles/harvest_r5.sh copies what is judged into profiles/r5/.
# GPU suite, smoke, the default bench line, rocprofv3 --kernel-trace --stats of the same command, and for configs 2 - 5 the L2's
# memory-side requests by size (two --pmc passes each, never combined with other trace domains).
set -u
root=$PWD; out=$root/gpurun_out/final5; mkdir -p $out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -12 ) > $out/gputest.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 ) > $out/smoke.log 2>&1
( timeout 1200 python bench.py 2>$out/bench.err | tail -1 ) > $out/bench.log
cd /tmp
cmd2="python $root/bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-other-configs"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats -o stats --output-format csv -- $cmd2 > $out/stats.log 2>&1
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/stats
req() { # tag kernel cmd...
  tag=$1; kern=$2; shift 2
  i=0
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp -d $out/req_${tag}_$i -o pmc --output-format csv -- "$@" > $out/req_${tag}_$i.log 2>&1
    find $out/req_${tag}_$i -name '*counter_collection.csv' -exec sh -c 'grep -E "Counter_Name|$3" "$1" > "$2"' _ {} $out/req_${tag}_$i.csv "$kern" \;
    rm -rf $out/req_${tag}_$i
  done
}
req cfg2 k_inflate_batch $cmd2
req cfg3 k_inflate_batch python $root/bench.py --config 3 --steps 1 --warmup 1 --no-legs --no-cpu-baseline
req cfg4 k_lzma python $root/bench.py --config 4 --steps 1 --warmup 0 --no-legs --no-cpu-baseline
req cfg5 k_deflate_batch python $root/bench.py --config 5 --steps 1 --warmup 1 --no-legs --no-cpu-baseline
cd $root
cat $out/gputest.log $out/smoke.log; head -c 600 $out/bench.log; echo; head -5 $out/kernel_stats.csv; ls $out

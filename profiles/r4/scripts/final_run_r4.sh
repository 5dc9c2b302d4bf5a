#!/bin/bash
# Round 4: everything the round's evidence consists of, in one gpurun call (from the repo root):
#   gpurun --timeout 3000 -- 'bash profiles/final_run_r4.sh'
# -> gpurun_out/final*/ ; what is judged is copied into profiles/r4/ by profiles/harvest_r4.sh afterwards.
set -u
mkdir -p gpurun_out/final
python -c "import torch" 2>/dev/null
( timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/final/gputest.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > gpurun_out/final/smoke.log 2>&1
# the default bench line (config 2 + other_configs + legs + cpu_baseline), then --stats and the two --pmc passes of config 2
bash profiles/collect.sh final all > gpurun_out/final/collect.log 2>&1
# HBM counters of configs 3, 4, 5 on HEAD
bash profiles/traffic_run.sh "3 4 5" > gpurun_out/final/traffic_run.log 2>&1
# known byte counts under the same counters
export TMPDIR=/tmp
root=$PWD
( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do lc=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/final/cal_$lc -o pmc --output-format csv -- python $root/profiles/pmc_calibrate.py > $root/gpurun_out/final/cal_$lc.log 2>&1
    find $root/gpurun_out/final/cal_$lc -name '*counter_collection.csv' -exec cp {} $root/gpurun_out/final/cal_$lc.csv \;
    rm -rf $root/gpurun_out/final/cal_$lc
  done )
bash profiles/collect.sh final_sq SQ > gpurun_out/final/collect_sq.log 2>&1
( AB_TAGS="base final prof final_prof" bash profiles/ab_k1_residency.sh; for tag in base final; do echo "== $tag small entries"; MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_$tag/libmzhip.so timeout 60 python tests/perf_probe.py 512 200000 8192 2>&1 | grep -v '^rep [01]\|amdgpu.ids'; done ) > gpurun_out/final/ab_k1_records.log 2>&1
( timeout 600 python tests/fuzz_gpu.py 12000 7 2>&1 | tail -3 ) > gpurun_out/final/fuzz_gpu.log 2>&1
( timeout 300 python bench.py --config 2 --entries 12500 --steps 10 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | tail -1 ) > gpurun_out/final/bench_shard12500.log 2>&1
cat gpurun_out/final/gputest.log gpurun_out/final/smoke.log gpurun_out/final/fuzz_gpu.log; tail -c 1500 gpurun_out/final/bench.log; ls gpurun_out/final gpurun_out/traffic_cfg* 2>/dev/null | head -60

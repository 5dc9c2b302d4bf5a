#!/bin/bash
# Everything the round's evidence consists of, in one gpurun call (from the repo root):
#   gpurun --timeout 2400 -- 'bash profiles/final_run.sh'
# -> gpurun_out/final*/ ; what is judged is copied into profiles/<round>/ afterwards (profiles/r3/README.md).
set -u
mkdir -p gpurun_out/final
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/final/gputest.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > gpurun_out/final/smoke.log 2>&1
bash profiles/collect.sh final all > gpurun_out/final/collect.log 2>&1
for c in 3 4 5; do timeout 600 python bench.py --config $c > gpurun_out/final/bench_cfg$c.log 2> gpurun_out/final/bench_cfg$c.err; done
bash profiles/collect.sh final_sq SQ > gpurun_out/final/collect_sq.log 2>&1
MZ_SQ_KERNEL=k_lzma_slot_batch MZ_SQ_CMD="python $PWD/bench.py --config 4 --entries 4608 --steps 1 --warmup 0 --no-cpu-baseline" bash profiles/collect.sh final_sq_k3 SQ > gpurun_out/final/collect_sq_k3.log 2>&1
MZ_SQ_KERNEL=k_deflate_batch MZ_SQ_CMD="python $PWD/bench.py --config 5 --entries 20000 --steps 1 --warmup 0 --no-cpu-baseline" bash profiles/collect.sh final_sq_k4 SQ > gpurun_out/final/collect_sq_k4.log 2>&1
( timeout 200 python tests/perf_codecs.py deflate_levels 2>&1 | grep -v amdgpu.ids ) > gpurun_out/final/k4_levels.log 2>&1   # GiB/s and ratio per DEFLATE class
[ -f minizip-ng_amd/_build_ab_k4prof/libmzhip.so ] && ( MZHIP_LIB=$PWD/minizip-ng_amd/_build_ab_k4prof/libmzhip.so timeout 200 python tests/perf_codecs.py deflate_levels 2>&1 | grep -v amdgpu.ids ) > gpurun_out/final/k4_levels_sections.log 2>&1   # make PROF=1 OUT=../_build_ab_k4prof
( timeout 600 python tests/fuzz_gpu.py 12000 7 2>&1 | tail -3 ) > gpurun_out/final/fuzz_gpu.log 2>&1
cat gpurun_out/final/gputest.log gpurun_out/final/smoke.log gpurun_out/final/fuzz_gpu.log; tail -2 gpurun_out/final/bench.log; for c in 3 4 5; do tail -1 gpurun_out/final/bench_cfg$c.log; done

#!/bin/bash
# Round 5, call 19: the T-reader leg with the archive's pages given back as the prime's upload passes them (mzhip_prime_uploaded +
# madvise on the application's side, integration/extract_threads.c) against the same with the pages kept; config 3 on 524 288 streams
set -u
root=$PWD; out=$root/gpurun_out/c19; mkdir -p $out
( MZ_NEAR=16 MZ_MODES=2 MZDROP_TRACE=1 timeout 300 python tests/perf_threads.py 2>&1 | grep -v amdgpu.ids | grep "mode\|returning\|readers done" | tail -40 ) > $out/threads_drop.log 2>&1
( MZDROP_KEEP_PAGES=1 MZ_NEAR=16 MZ_MODES=2 MZDROP_TRACE=1 timeout 300 python tests/perf_threads.py 2>&1 | grep -v amdgpu.ids | grep "mode\|returning\|readers done" | tail -40 ) > $out/threads_keep.log 2>&1
grep mode $out/threads_drop.log; echo; grep mode $out/threads_keep.log
( timeout 400 python bench.py --config 3 --no-cpu-baseline 2>$out/bench3.err | tail -1 ) > $out/bench3.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c19/bench3.log").read())
print("cfg3", d["value"], d["ms_per_step"], d["config"]["unique_streams"])
PY

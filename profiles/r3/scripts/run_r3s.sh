#!/bin/bash
set -u
o=gpurun_out/r3s; mkdir -p $o
( timeout 300 python -m pytest tests/test_gpu_prime.py tests/test_gpu_inflate.py -x -q 2>&1 | tail -3 ) > $o/tests.log 2>&1
( MZ_NEAR=16 MZ_MODES=2 timeout 100 python tests/perf_threads.py 2>&1 | grep "^mode\|bound" ) > $o/threads.log 2>&1
( MZ_NEAR=0 MZ_MODES=2 timeout 100 python tests/perf_threads.py 2>&1 | grep "^mode\|bound" ) >> $o/threads.log 2>&1
( timeout 400 python bench.py 2> $o/bench.err ) > $o/bench.log
cat $o/tests.log $o/threads.log; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3s/bench.log").read().strip().splitlines()[-1])
print(d["value"], d["config"].get("host_placement"), d["cpu_baseline"]["value"], {k:v for k,v in d["legs"].items() if not k.endswith("sample")})
PY

#!/bin/bash
# Round 5, call 12: the whole GPU suite on HEAD (C-level gather, 4-rank bench lines, window fuzz test, auto-prime default), config 3 on 524 288 unique streams
set -u
root=$PWD; out=$root/gpurun_out/c12; mkdir -p $out
( timeout 2400 python -X faulthandler -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -25 ) > $out/gputest.log 2>&1
cat $out/gputest.log
( timeout 600 python bench.py --config 3 --no-cpu-baseline 2>$out/bench3.err | tail -1 ) > $out/bench3.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c12/bench3.log").read())
print("cfg3", d["value"], d["ms_per_step"], d["config"]["unique_streams"], d["data"][:200])
PY
tail -2 $out/bench3.err

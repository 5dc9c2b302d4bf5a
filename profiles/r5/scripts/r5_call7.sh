#!/bin/bash
# Round 5, call 7: the whole GPU suite with auto-prime on by default, the collapsed host ABI and the ADVICE fixes; the default bench line
set -u
root=$PWD; out=$root/gpurun_out/c7; mkdir -p $out
( timeout 1800 python -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 ) > $out/gputest.log 2>&1
cat $out/gputest.log
( timeout 700 python bench.py 2>$out/bench.err | tail -1 ) > $out/bench.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c7/bench.log").read())
print("cfg2", d["value"], d["ms_per_step"], "legs", {k:v for k,v in d["legs"].items() if not k.endswith("_sample")})
print({k:(v["value"],v["ms_per_step"]) for k,v in d["other_configs"].items()})
print("cpu 1-thread:", d["cpu_baseline"]["sample"][-30:])
PY
tail -3 $out/bench.err

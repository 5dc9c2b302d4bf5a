#!/bin/bash
# Round 5, call 17: the default bench line on HEAD (config 2 with legs and the other configs behind it)
set -u
root=$PWD; out=$root/gpurun_out/c17; mkdir -p $out
( timeout 1500 python bench.py 2>$out/bench.err | tail -1 ) > $out/bench.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c17/bench.log").read())
print("cfg2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
for k,v in (d.get("legs") or {}).items(): print("leg", k, v if not isinstance(v, dict) else {a:b for a,b in v.items() if a in ("value","unit","gib_s","seconds")})
for k,v in (d.get("other_configs") or {}).items(): print("cfg", k, v.get("value"), v.get("unit"), v.get("config",{}).get("unique_streams"))
print("cpu", d.get("cpu_baseline"))
PY
tail -3 $out/bench.err

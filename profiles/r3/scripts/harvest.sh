#!/bin/bash
# Copy what profiles/final_run.sh produced under gpurun_out/ into profiles/<round>/ (the judged, tracked copies):
#   bash profiles/harvest.sh r3
set -u
r=${1:-r3}
d=profiles/$r
mkdir -p $d
cp gpurun_out/final/kernel_stats.csv $d/kernel_stats_final.csv
cp gpurun_out/final/pmc_fetch_size.csv gpurun_out/final/pmc_write_size.csv $d/
cp gpurun_out/final/bench.log $d/bench_cfg2.log
for c in 3 4 5; do cp gpurun_out/final/bench_cfg$c.log $d/bench_cfg$c.log; done
cp gpurun_out/final/gputest.log $d/gputest_final.log
cp gpurun_out/final/fuzz_gpu.log $d/fuzz_gpu_final.log
cp gpurun_out/final/k4_levels.log gpurun_out/final/k4_levels_sections.log $d/ 2>/dev/null
for t in final_sq:k1 final_sq_k3:k3 final_sq_k4:k4; do src=${t%%:*}; k=${t##*:}; for i in 1 2 3; do cp gpurun_out/$src/pmc_sq$i.csv $d/pmc_sq${i}_${k}_final.csv; done; done
python3 - "$d" <<'PY'
import csv, json, sys
d = sys.argv[1]
def mean(f):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))]
    return sum(v) / len(v) * 1024
p = d + "/hbm_traffic.json"
t = json.load(open(p))
t["fetch_bytes_per_launch"] = int(mean(d + "/pmc_fetch_size.csv"))
t["write_bytes_per_launch"] = int(mean(d + "/pmc_write_size.csv"))
b = json.loads(open(d + "/bench_cfg2.log").read().strip().splitlines()[-1])
t["algorithmic_bytes_per_launch"] = b["roofline"]["algorithmic_bytes_per_launch"]
t["kernel_ms_per_launch"] = b["roofline"]["kernel_ms"]
import subprocess
t["commit"] = subprocess.run(["git", "log", "-1", "--format=%h"], capture_output=True, text=True).stdout.strip()
json.dump(t, open(p, "w"), indent=1)
print(t["fetch_bytes_per_launch"], t["write_bytes_per_launch"], t["kernel_ms_per_launch"])
PY
python3 profiles/resource_usage.py > $d/kernel_resource_usage.txt   # (compiler remarks of the default build at HEAD: no GPU needed)
head -2 $d/kernel_stats_final.csv | cut -c1-120

#!/bin/bash
# gpurun_out/final5b/ -> profiles/r5/ (run here, in the repository: the JSONs carry the commit)
set -u
src=gpurun_out/final5b; dst=profiles/r5
cp $src/gputest.log $dst/gputest_final.log; cp $src/smoke.log $dst/smoke_final.log; cp $src/bench.log $dst/bench_final.log
cp $src/kernel_stats.csv $dst/kernel_stats.csv
for c in 2 3; do cp $src/req_cfg${c}_1.csv $dst/req_final_cfg${c}_1.csv; cp $src/req_cfg${c}_2.csv $dst/req_final_cfg${c}_2.csv; done
python - <<'PY'
import json, shutil, subprocess, sys
d = json.loads(open("gpurun_out/final5b/bench.log").read())
alg = {2: (d["roofline"]["algorithmic_bytes_per_launch"], d["config"]["entries_total"], d["config"]["entry_bytes"], "k_inflate_batch"),
       3: (d["other_configs"]["3"]["roofline"]["algorithmic_bytes_per_launch"], 1000000, 8192, "k_inflate_batch")}
for c, (a, n, sz, kn) in sorted(alg.items()):
    for i in (1, 2): shutil.copy("gpurun_out/final5b/req_cfg%d_%d.csv" % (c, i), "/tmp/req_f%d_%d.csv" % (c, i))
    subprocess.run([sys.executable, "profiles/req_harvest.py", "r5", "/tmp", "f%d" % c, str(c), str(n), str(sz), str(a), kn], check=True)
    print("config", c, "harvested")
PY
python profiles/resource_usage.py > $dst/kernel_resource_usage.txt 2>&1
ls $dst | wc -l

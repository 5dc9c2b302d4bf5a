#!/bin/bash
# gpurun_out/final5/ -> profiles/r5/ (run here, in the repository: the JSONs carry the commit)
set -u
src=gpurun_out/final5; dst=profiles/r5
cp $src/gputest.log $dst/gputest_final.log; cp $src/smoke.log $dst/smoke_final.log; cp $src/bench.log $dst/bench_final.log
cp $src/kernel_stats.csv $dst/kernel_stats.csv
for c in 2 3 4 5; do cp $src/req_cfg${c}_1.csv $dst/req_final_cfg${c}_1.csv; cp $src/req_cfg${c}_2.csv $dst/req_final_cfg${c}_2.csv; done
python - <<'PY'
import json, subprocess, sys
d = json.loads(open("gpurun_out/final5/bench.log").read())
alg = {2: (d["roofline"]["algorithmic_bytes_per_launch"], d["config"]["entries_total"], d["config"]["entry_bytes"], "k_inflate_batch")}
kern = {3: "k_inflate_batch", 4: "k_lzma+", 5: "k_deflate_batch"}
size = {3: 8192, 4: 1048576, 5: 65536}; ents = {3: 1000000, 4: 10000, 5: 100000}
for k, v in d["other_configs"].items():
    alg[int(k)] = (v["roofline"]["algorithmic_bytes_per_launch"], ents[int(k)], size[int(k)], kern[int(k)])
for c, (a, n, sz, kn) in sorted(alg.items()):
    import os, shutil
    for i in (1, 2): shutil.copy("gpurun_out/final5/req_cfg%d_%d.csv" % (c, i), "/tmp/req_f%d_%d.csv" % (c, i))
    subprocess.run([sys.executable, "profiles/req_harvest.py", "r5", "/tmp", "f%d" % c, str(c), str(n), str(sz), str(a), kn], check=True)
    print("config", c, "harvested")
PY
python profiles/resource_usage.py > $dst/kernel_resource_usage.txt 2>&1
ls $dst | head -80

#!/usr/bin/env python3
"""gpurun_out/traffic_cfgN/pmc_{fetch,write}_size.csv (profiles/traffic_run.sh) -> profiles/<round>/hbm_traffic_cfgN.json + the CSVs.
    python profiles/traffic_harvest.py r3 3 4"""
import csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rnd, cfgs = sys.argv[1], [int(c) for c in sys.argv[2:]]
import bench  # noqa: E402  (CONFIGS: workload text, entries, entry size)

LAUNCHES = 4  # bench.py --steps 3 --warmup 1
for c in cfgs:
    src = os.path.join(ROOT, "gpurun_out", "traffic_cfg%d" % c)
    out = {}
    for what in ("fetch", "write"):
        f = os.path.join(src, "pmc_%s_size.csv" % what)
        rows = list(csv.DictReader(open(f)))
        kernels = sorted({r["Kernel_Name"].split("(")[0] for r in rows})
        out[what] = int(sum(float(r["Counter_Value"]) for r in rows) * 1024 / LAUNCHES)  # every kernel of a step, per step
        shutil.copy(f, os.path.join(ROOT, "profiles", rnd, "pmc_%s_size_cfg%d.csv" % (what, c)))
    cfg = bench.CONFIGS[c]
    t = {"kernel": " + ".join(kernels), "workload": cfg["workload"] % (cfg["entries"], cfg["size"]), "entries": cfg["entries"],
         "entry_bytes": cfg["size"], "fetch_bytes_per_launch": out["fetch"], "write_bytes_per_launch": out["write"],
         "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --config %d --steps 3 "
                   "--warmup 1 --no-legs --no-cpu-baseline` (profiles/traffic_run.sh), counter value x 1024 B, all kernels of a step "
                   "summed, mean of the %d steps (pmc_*_size_cfg%d.csv)" % (c, LAUNCHES, c),
         "corrections": "none applied (see hbm_traffic.json)",
         "commit": subprocess.run(["git", "log", "-1", "--format=%h"], capture_output=True, text=True, cwd=ROOT).stdout.strip()}
    json.dump(t, open(os.path.join(ROOT, "profiles", rnd, "hbm_traffic_cfg%d.json" % c), "w"), indent=1)
    print(c, t["kernel"], out)

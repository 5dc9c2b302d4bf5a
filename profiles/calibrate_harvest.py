#!/usr/bin/env python3
"""gpurun_out/<dir>/cal_{fetch,write}.csv (profiles/pmc_calibrate.py under rocprofv3 --pmc) -> profiles/<round>/pmc_calibration.json:
counter value x 1024 B against the known byte count of each calibration kernel.
    python profiles/calibrate_harvest.py r4 gpurun_out/final"""
import csv, json, os, sys

rnd, src = sys.argv[1], sys.argv[2]
N = 2 << 30
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"bytes_per_kernel": N, "note": "counter value x 1024 B (rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB) divided by the bytes the kernel is known to move; "
       "last launch of each kernel (the first warms the clocks)"}
for what in ("fetch", "write"):
    rows = list(csv.DictReader(open(os.path.join(src, "cal_%s.csv" % what))))
    R, W = (N, 0) if what == "fetch" else (0, N)
    for key, pat, expect in (("crc32_16B_loads", "k_crc32_batch", R), ("torch_fill_u8_vec16", "FillFunctor<unsigned char>", W),
                             ("torch_add_u8_vec16", "add<unsigned char>", N),
                             ("read_1B_per_lane", "k_cal_read<unsigned char>", R), ("read_4B_per_lane", "k_cal_read<unsigned int>", R),
                             ("read_16B_per_lane", "k_cal_read16", R), ("write_1B_per_lane", "k_cal_write<unsigned char>", W),
                             ("write_4B_per_lane", "k_cal_write<unsigned int>", W), ("write_16B_per_lane", "k_cal_write16", W)):
        v = [float(r["Counter_Value"]) * 1024 for r in rows if pat in r["Kernel_Name"]]
        if not v:
            continue
        out.setdefault(key, {})[what + "_counter_bytes"] = int(v[-1])
        out[key][what + "_known_bytes"] = expect
        if expect:
            out[key][what + "_counter_over_known"] = round(v[-1] / expect, 4)
json.dump(out, open(os.path.join(ROOT, "profiles", rnd, "pmc_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

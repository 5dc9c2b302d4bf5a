#!/usr/bin/env python3
"""The L2's memory-side requests of one kernel -> profiles/<round>/hbm_traffic*.json.

    python profiles/req_harvest.py <round> <dir with req_<tag>_{1,2}.csv> <tag> <config> <entries> <entry_bytes> <algorithmic_bytes> [kernel] [ablation tags ...]

req_<tag>_1.csv: TCC_EA0_RDREQ_{sum,32B_sum,64B_sum,128B_sum}; req_<tag>_2.csv: TCC_EA0_WRREQ_{sum,64B_sum}, TCC_HIT_sum,
TCC_MISS_sum -- separate rocprofv3 --pmc passes of the same command (profiles/gpu.sh req / probe / evidence; round 5: profiles/r5/scripts/), rows of the named kernel only.
Bytes read = 32 n32 + 64 n64 + 128 n128 (what FETCH_SIZE is derived from, with the 128-byte requests counted as 128 and not
as 64: the guide's "FETCH_SIZE is half the bytes" and profiles/r4/pmc_calibration.json are this); bytes written = 32 (n -
n64) + 64 n64 (= WRITE_SIZE).  `traffic_raw` keeps what the two derived counters would have said.  Ablation tags (builds
with one stage removed, wrong output) give the table "bytes by section" = HEAD minus the ablated build."""
import csv, collections, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(d, tag, kernel):
    """kernel = a name prefix; with a trailing '+' the rows of every matching kernel are ADDED (one step of a workload that is
    several launches: config 4 = k_lzma_slot_batch + k_lzma_batch over what it gives back), else they are averaged (the same
    kernel launched several times)"""
    out = {}
    add = kernel.endswith("+")
    kernel = kernel.rstrip("+")
    for i in (1, 2):
        rows = [r for r in csv.DictReader(open(os.path.join(d, "req_%s_%d.csv" % (tag, i)))) if kernel in r["Kernel_Name"]]  # (a substring: k_inflate_batch is "void k_inflate_batch<false>(InflateArgs)" since it became a template)
        acc, dur, per_kernel = collections.defaultdict(list), {}, collections.Counter()
        for r in rows:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in dur:
                per_kernel[r["Kernel_Name"]] += 1
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        # '+': one STEP of the workload is one launch of each matching kernel; the command ran `steps` of them (the kernel launched
        # most often says how many) -- the sums are per step
        steps = max(per_kernel.values()) if (add and per_kernel) else 1
        for k, v in acc.items():
            out[k] = sum(v) / steps if add else sum(v) / len(v)
        out["launches"] = steps if add else len(dur)
        out["kernel_ms_under_pmc"] = round(sum(dur.values()) / (steps if add else max(1, len(dur))), 3)
    n32, n64, n128 = out.get("TCC_EA0_RDREQ_32B_sum", 0), out.get("TCC_EA0_RDREQ_64B_sum", 0), out.get("TCC_EA0_RDREQ_128B_sum", 0)
    w, w64 = out.get("TCC_EA0_WRREQ_sum", 0), out.get("TCC_EA0_WRREQ_64B_sum", 0)
    rd = 32 * n32 + 64 * n64 + 128 * n128
    wr = 32 * (w - w64) + 64 * w64
    fetch_raw = 64 * (out.get("TCC_EA0_RDREQ_sum", 0) - n32) + 32 * n32  # FETCH_SIZE x 1024 without its BUBBLE term
    return dict(read_bytes_per_launch=int(rd), write_bytes_per_launch=int(wr), read_requests={"32B": int(n32), "64B": int(n64), "128B": int(n128)},
                write_requests={"32B": int(w - w64), "64B": int(w64)}, tcc_hit=int(out.get("TCC_HIT_sum", 0)), tcc_miss=int(out.get("TCC_MISS_sum", 0)),
                traffic_raw=int(fetch_raw + wr), launches=out["launches"], kernel_ms_under_pmc=out["kernel_ms_under_pmc"])


def kernel_src_hash():
    """what names the kernels a measurement was taken on: a hash of every device source (minizip-ng_amd/csrc/*.h, *.inc, *.hip).  bench.py
    computes the same and says in `traffic_source` whether the counters belong to the build it is timing."""
    import glob
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "minizip-ng_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "*.hip"))):
        if os.path.basename(f) in ("mzhip_prime.inc", "shim_common.h"):  # the host-side prime cache and the shims' glue: nothing a kernel or its launch sees
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    rnd, d, tag, cfg, entries, entry_bytes, alg = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    kernel = sys.argv[8] if len(sys.argv) > 8 else "k_inflate_batch"
    t = load(d, tag, kernel)
    t.update(kernel=kernel, config=cfg, entries=entries, entry_bytes=entry_bytes, algorithmic_bytes_per_launch=alg,
             traffic_over_algorithmic=round((t["read_bytes_per_launch"] + t["write_bytes_per_launch"]) / alg, 2),
             source="rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_{sum,32B,64B,128B}_sum | TCC_EA0_WRREQ_{sum,64B}_sum TCC_HIT_sum TCC_MISS_sum in two passes "
                    "of one command on one binary; bytes = sum of size x count, mean of the kernel's launches",
             commit=subprocess.run(["git", "log", "-1", "--format=%h"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or os.environ.get("HARVEST_COMMIT", "unknown"),
             kernel_src_hash=kernel_src_hash())
    sections = {}
    for ab in sys.argv[9:]:
        name, atag = ab.split("=")
        a = load(d, atag, kernel)
        sections[name] = dict(read_bytes=t["read_bytes_per_launch"] - a["read_bytes_per_launch"], write_bytes=t["write_bytes_per_launch"] - a["write_bytes_per_launch"],
                              kernel_ms_under_pmc=a["kernel_ms_under_pmc"], build=atag)
    if sections:
        t["bytes_by_section"] = sections
        t["bytes_by_section_note"] = ("HEAD minus a build with that stage removed (MZ_ABLATE / MZ_CHASE_X: wrong output, traffic and time only); the stages "
                                      "share the L2, so the differences overlap and do not add up to the total")
    name = "hbm_traffic.json" if cfg == 2 and entries == 100000 else "hbm_traffic_cfg%d_%dx%d.json" % (cfg, entries, entry_bytes) if cfg == 2 else "hbm_traffic_cfg%d.json" % cfg
    path = os.path.join(os.environ.get("HARVEST_OUT") or os.path.join(ROOT, "profiles", rnd), name)  # (HARVEST_OUT: on the GPU box only gpurun_out/ comes back)
    json.dump(t, open(path, "w"), indent=1)
    print(path, "read %.2f GB write %.2f GB = %.2f x algorithmic (raw counters: %.2f x)" % (t["read_bytes_per_launch"] / 1e9, t["write_bytes_per_launch"] / 1e9,
          t["traffic_over_algorithmic"], t["traffic_raw"] / alg))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the default build, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage).  No GPU needed.

    python profiles/resource_usage.py > profiles/r2/kernel_resource_usage.txt
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "minizip-ng_amd", "csrc", "mzhip_kernels.hip")


def make_flags():
    """The -D knobs of the default `make` (so the table describes the library that ships)."""
    out = subprocess.run(["make", "-C", os.path.dirname(SRC), "-n", "-B"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if "hipcc" in line and "mzhip_kernels.hip" in line:
            return [t for t in line.split() if t.startswith("-D") or t.startswith("-O") or t.startswith("-std")]
    return ["-O3"]


def short(name):
    m = re.match(r"_Z(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    t = re.match(r"ILi(\d+)E", rest)
    b = re.match(r"ILb([01])E", rest)  # (k_inflate_batch<bool RESUMABLE>)
    return base + ("<%s>" % t.group(1) if t else "<%s>" % ("true" if b.group(1) == "1" else "false") if b else "")


def main():
    flags = make_flags()
    cmd = ["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-c", "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", "-I" + os.path.join(ROOT, "include")] + flags + [SRC]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = dict(name=short(m.group(1)))
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    print("hipcc --offload-arch=gfx950 %s -Rpass-analysis=kernel-resource-usage on minizip-ng_amd/csrc/mzhip_kernels.hip"
          % " ".join(f for f in flags if not f.startswith("-std")))
    print("(sSpill = SGPRs parked in VGPR lanes (v_writelane), not memory; scratch = bytes per lane; lds = static bytes per block)")
    print("%-30s %5s %5s %7s %5s %6s %6s %7s" % ("kernel", "SGPR", "VGPR", "scratch", "occ", "sSpill", "vSpill", "lds"))
    seen = set()
    for r in rows:
        if r["name"] in seen or not r["name"].startswith("k_"):
            continue
        seen.add(r["name"])
        print("%-30s %5d %5d %7d %5d %6d %6d %7d" % (r["name"], r.get("sgpr", 0), r.get("vgpr", 0), r.get("scratch", 0),
                                                      r.get("occ", 0), r.get("sspill", 0), r.get("vspill", 0), r.get("lds", 0)))
    return 0


if __name__ == "__main__":
    sys.exit(main())

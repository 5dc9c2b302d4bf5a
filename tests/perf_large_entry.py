#!/usr/bin/env python3
"""One large DEFLATE entry through the drop-in (window mode, a wave per block): where the time goes.
    python tests/perf_large_entry.py [sparse|text|mixed] [GiB] [level]
Writes a ZIP64 archive with one entry of that size, reads it with the all-reference reader and with libmzhipdrop.so in a
child process with MZHIP_STREAM_STATS=1 / MZHIP_PAR_TRACE=1 (the shim's and the device call's own clocks), prints both times,
the stats line and the first and last trace lines."""
import json
import os
import subprocess
import sys
import tempfile
import time
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "text"
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
DROP = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
total = int(gib * (1 << 30)) + 12345
text = synth.bench_corpus()[0]
piece = {"sparse": (b"sparse " * 1024 + bytes(120000)) * 8, "text": text * 2,
         "mixed": text + os.urandom(200000) + bytes(300000) + text[::-1]}[kind]
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "big.zip")
    t0 = time.time()
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED, allowZip64=True, compresslevel=level) as zf:
        with zf.open("huge.bin", "w", force_zip64=True) as f:
            left = total
            while left > 0:
                k = min(left, len(piece))
                f.write(piece[:k])
                left -= k
    print("%s, %.2f GiB at level %d: archive %.1f MiB, written in %.1f s" % (kind, total / (1 << 30), level, os.path.getsize(path) / (1 << 20), time.time() - t0))
    ref = oracle.ref()
    table = ref.zip_index(path)
    cd = table[:, 6].copy()
    sec_r, crc_r, ulen_r, st_r = ref.zip_read_all(path, cd, nthreads=1, own_crc=False)
    assert (st_r == 0).all() and int(ulen_r[0]) == total
    prog = ("import sys, json\nsys.path.insert(0, %r)\nimport numpy as np, oracle\nhip = oracle.MzDriver(%r)\ncd = np.array(%r, dtype=np.int64)\n"
            "for rep in range(2):\n    sec, crc, ulen, st = hip.zip_read_all(%r, cd, nthreads=1, own_crc=False)\n"
            "    print(json.dumps(dict(sec=sec, crc=[int(x) for x in crc], ulen=[int(x) for x in ulen], st=[int(x) for x in st])))\n"
            % (ROOT, DROP, [int(x) for x in cd], path))
    for par in (("1",) if os.environ.get("MZ_PERF_SKIP_OFF") else ("1", "0")):
        env = dict(os.environ, MZHIP_STREAM_STATS="1", MZHIP_PAR_TRACE="1", MZHIP_STREAM_PARALLEL=par)
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        runs = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        for g in runs:
            assert g["st"] == [0] and g["ulen"] == [int(ulen_r[0])] and g["crc"] == [int(crc_r[0])]
        print("many-wave decode %s: %.2f s / %.2f s (first / second read in the process) = %.2f GiB/s; the all-reference reader: %.2f s = %.2f GiB/s"
              % ("on" if par == "1" else "OFF", runs[0]["sec"], runs[1]["sec"], total / runs[1]["sec"] / (1 << 30), sec_r, total / sec_r / (1 << 30)))
        lines = [l for l in r.stderr.splitlines() if l.startswith("mzhip")]
        tr = [l for l in lines if "parallel window" in l]
        for l in tr[:3] + (["..."] if len(tr) > 6 else []) + tr[-3:]:
            print("   ", l)
        for l in lines:
            if "window mode" in l:
                print("   ", l)

#!/usr/bin/env python3
"""One large entry WRITTEN through the drop-in (mz_stream_zlib / mz_stream_lzma WRITE in segments) and through the all-reference
build, then read back by the reference:   python tests/perf_large_write.py [GiB=1] [method=8] [level=1]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
method = int(sys.argv[2]) if len(sys.argv) > 2 else 8
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
total = int(gib * (1 << 30)) + 12345
piece = np.frombuffer(synth.bench_corpus()[0] * 2, dtype=np.uint8)
hip, ref = oracle.MzDriver(os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")), oracle.ref()
with tempfile.TemporaryDirectory() as tmp:
    res = {}
    for name, drv, reps in (("drop-in", hip, 2), ("reference", ref, 1)):
        if name == "reference" and (method != 8 or level > 1) and gib > 0.3:
            total_r = total // 8  # (liblzma / zlib-6 on one thread: a slice is enough for a rate)
        else:
            total_r = total
        for rep in range(reps):
            path = os.path.join(tmp, "%s.zip" % name)
            t0 = time.time()
            drv.zip_write_repeat(path, piece, total_r, method=method, level=level)
            dt = time.time() - t0
            res[name] = (dt, total_r, os.path.getsize(path))
            print("%s: %.2f GiB, method %d level %d written in %.2f s = %.2f GiB/s, archive %.1f MiB (ratio %.3f)" % (
                name, total_r / (1 << 30), method, level, dt, total_r / (1 << 30) / dt, os.path.getsize(path) / (1 << 20), os.path.getsize(path) / total_r), flush=True)
    # the reference reads what the drop-in wrote
    path = os.path.join(tmp, "drop-in.zip")
    table = ref.zip_index(path)
    _, crc, ulen, st = ref.zip_read_all(path, table[:, 6].copy(), nthreads=1, own_crc=True)
    print("the all-reference reader on the drop-in's archive: statuses %s, sizes %s" % (st.tolist(), ulen.tolist()))
    assert (st == 0).all() and int(ulen[0]) == total

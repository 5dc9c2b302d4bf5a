"""Entries WRITTEN one by one through the unmodified mz_zip writer (mz_zip_writer_add_buffer: open, write, close per entry) on the
drop-in and on the all-reference build: what an entry costs when nothing was primed (mzhip_prime_write).
    python tests/perf_write_entries.py [n size [drop-in only]]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
import oracle
from tests import synth
hip, ref = oracle.MzDriver("integration/_build/libmzhipdrop.so"), oracle.ref()
c = np.frombuffer(synth.corpus(), dtype=np.uint8)
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(2000, 65536), (500, 1 << 20), (64, 8 << 20)]
only_drop_in = len(sys.argv) > 3
for n, size in shapes:
    rnd = np.random.RandomState(1)
    offs = rnd.randint(0, max(1, len(c) - min(size, len(c) // 2)), size=n).astype(np.int64)
    blob = c
    if size > len(c) // 2:
        blob = np.tile(c, size // len(c) + 2)
        offs = rnd.randint(0, len(blob) - size, size=n).astype(np.int64)
    lens = np.full(n, size, dtype=np.int32)
    with tempfile.TemporaryDirectory() as tmp:
        for name, drv in ((("drop-in", hip),) if only_drop_in else (("drop-in", hip), ("reference", ref))):
            for level in (1, 6):
                p = os.path.join(tmp, "w.zip")
                t0 = time.time(); drv.zip_write(p, blob, offs, lens, method=8, level=level); dt = time.time() - t0
                print("%-9s %5d x %8d B level %d: %.2f s = %.3f GiB/s (%.0f us per entry), ratio %.3f" % (name, n, size, level, dt, n * size / 2**30 / dt, dt / n * 1e6, os.path.getsize(p) / (n * size)), flush=True)

"""Entries READ one by one through the unmodified mz_zip reader on the drop-in with nothing primed (MZHIP_AUTOPRIME=0: the per-entry
path every look-up that misses takes) and on the all-reference build.   python tests/perf_read_entries.py [n size [drop-in only]]
MZ_PERF_PYZIP=1: the archive is written by Python's zipfile (sizes in the local headers) instead of the reference's writer."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["MZHIP_AUTOPRIME"] = "0"
import oracle
from tests import synth
hip, ref = oracle.MzDriver("integration/_build/libmzhipdrop.so"), oracle.ref()
c = np.frombuffer(synth.corpus(), dtype=np.uint8)
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(2000, 65536), (2000, 8192), (300, 1 << 20)]
for n, size in shapes:
    rnd = np.random.RandomState(1)
    blob = c if size <= len(c) // 2 else np.tile(c, size // len(c) + 2)
    offs = rnd.randint(0, len(blob) - size, size=n).astype(np.int64)
    lens = np.full(n, size, dtype=np.int32)
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "r.zip")
        if os.environ.get("MZ_PERF_PYZIP") == "1":  # sizes in the local headers (Info-ZIP, 7-Zip, Python ... on a seekable output); the reference's writer leaves them to a data descriptor
            import zipfile
            with zipfile.ZipFile(p, "w", zipfile.ZIP_DEFLATED, compresslevel=6) as zf:
                for i in range(n):
                    zf.writestr("e/%06d" % i, blob[int(offs[i]):int(offs[i]) + size].tobytes())
        else:
            ref.zip_write(p, blob, offs, lens, method=8, level=6)
        cd = ref.zip_index(p)[:, 6].copy()
        for name, drv in ((("drop-in", hip),) if len(sys.argv) > 3 else (("drop-in", hip), ("reference", ref))):
            sec, crc, ulen, st = drv.zip_read_all(p, cd, nthreads=1, own_crc=False)
            assert (st == 0).all() and (ulen == size).all()
            print("%-9s %5d x %8d B: %.2f s = %.3f GiB/s (%.0f us per entry)" % (name, n, size, sec, n * size / 2**30 / sec, sec / n * 1e6), flush=True)

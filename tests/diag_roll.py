"""Diagnostic (not a pytest): the rolling auto-prime in a loop on the device, every non-zero status with mzhip_last_error()."""
import ctypes as C
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import test_autoprime_emul as A  # noqa: E402

mz = importlib.import_module("minizip-ng_amd")
mz.require_gpu()
DROP = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
hip, ref, L = oracle.MzDriver(DROP), oracle.ref(), A.bind(mz.lib())
L.mzhip_last_error.restype = C.c_char_p
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ["MZHIP_AUTOPRIME"] = sys.argv[3] if len(sys.argv) > 3 else "256k"
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "roll.zip")
    lens = A.make_archive(ref, path, 420, 16384, seed=7)
    cd = ref.zip_index(path)[:, 6].copy()
    out_off = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
    o_ref = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
    _, crc_r, ulen_r, st_r = ref.zip_read_all(path, cd, nthreads=1, own_crc=False, out=o_ref, out_off=out_off)
    bad = 0
    for rep in range(reps):
        o_hip = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
        _, crc_h, ulen_h, st_h = hip.zip_read_all(path, cd, nthreads=nthreads if rep % 2 == 0 else 1, own_crc=False, out=o_hip, out_off=out_off)
        s = A.stats(L)
        okb = bool((o_hip == o_ref).all())
        # (a look-up that misses is not an error: with more than one reader a window can be evicted under the one that was about to
        # use it, or given up by the thrash guard, and the entry takes the per-entry path -- a few per pass at most, or something leaks)
        if (st_h != 0).any() or not okb or s["misses"] > 8:
            bad += 1
            print("rep %d: statuses %s at %s, bytes equal %s, stats %s, last error %r" % (rep, st_h[st_h != 0], np.nonzero(st_h)[0], okb, s,
                                                                                        L.mzhip_last_error()), flush=True)
        if rep % 3 == 2:
            L.mzhip_prime_clear()
    print("diag_roll: %d of %d passes were not clean" % (bad, reps), flush=True)

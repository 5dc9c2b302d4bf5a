#!/usr/bin/env python3
"""GPU-box probe (not a pytest): tests/test_gpu_xz.py::test_xz_differential_fuzz_batch at any size -- N corrupted / truncated
.xz streams (variants of synth.xz_cases()) through mzhip_xz_batch in one launch against the oracle restatement (pinned to the
compiled reference on the same kind of variants by tests/fuzz_oracle_xz.py): the same accept / reject decision and error
class; on accept the same bytes, consumed input and CRC.
    python tests/fuzz_gpu_xz.py [N=20000] [seed=1]"""
import ctypes as C
import multiprocessing as mp
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

CAP = 200000
BASES = None


def init():
    global BASES
    BASES = [x for n, d, x in synth.xz_cases() if 0 < len(d) <= 100000 and not ("lp4" in n or "lc4" in n or "lc1lp3" in n)]  # (as the test: lc + lp = 4 is the one-buffer host path's)


def prepare(seed):
    rnd = random.Random(seed)
    x = bytearray(rnd.choice(BASES))
    k = rnd.randrange(5)
    if k == 0:
        x[rnd.randrange(len(x))] ^= 1 << rnd.randrange(8)
    elif k == 1:
        x[rnd.randrange(len(x))] = rnd.randrange(256)
    elif k == 2:
        del x[rnd.randrange(1, len(x)):]
    elif k == 3:
        x[rnd.randrange(min(len(x), 40))] = rnd.randrange(256)
    else:
        x[-rnd.randrange(1, 40)] = rnd.randrange(256)
    x = bytes(x)
    so, uo, oo = oracle.xz_decode(x, CAP)
    return x, so, uo, len(oo), zlib.crc32(oo)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    with mp.Pool(initializer=init) as pool:      # forks before the HIP context exists
        rows = pool.map(prepare, [seed * 1000003 + i for i in range(n)], chunksize=16)
    from tests import gpu_util
    from tests.test_gpu_xz import run_xz

    gpu_util.mz.require_gpu()
    L = gpu_util.mz.lib()
    L.mzhip_xz_batch.restype = C.c_int32
    L.mzhip_xz_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5
    pays = [r[0] for r in rows]
    b, h_out, out_len, in_used, crc, status = run_xz(gpu_util, pays, [CAP] * len(pays))
    bad = n_ok = 0
    for i, (x, so, uo, ol, k) in enumerate(rows):
        if so == 0:
            n_ok += 1
            ok = (status[i], in_used[i], out_len[i], int(crc[i])) == (0, uo, ol, k)
        elif so == -109:
            ok = status[i] in (-109, -3)
        else:
            ok = status[i] == so or (status[i], so) == (-109, -3)
        if not ok:
            bad += 1
            if bad < 10:
                print("MISMATCH", i, "gpu", status[i], in_used[i], out_len[i], hex(int(crc[i])), "oracle", so, uo, ol, hex(k), len(x))
    print("gpu xz fuzz: %d streams (%d decode) -- %d mismatches" % (len(rows), n_ok, bad))
    sys.exit(1 if bad else 0)

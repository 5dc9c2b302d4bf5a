#!/usr/bin/env python3
"""GPU-box probe (not a pytest): one large differential run of K3 through the C ABI (mzhip_lzma_batch) -- the streams of
tests/fuzz_oracle_lzma.py (random method-14 payloads of every preset and hand-set lc / lp / pb / dictionary size: whole, cut,
a byte flipped or replaced anywhere, properties and header included), one launch; every entry is checked on the host cores
against the oracle restatement (which that script pins to the compiled reference on the same streams): refused by one =
refused by the other; bytes, consumed input and CRC wherever the stream decodes.
    python tests/fuzz_gpu_lzma.py [N=3000] [seed=1]"""
import importlib.util
import multiprocessing as mp
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import oracle  # noqa: E402

_spec = importlib.util.spec_from_file_location("fuzz_oracle_lzma", os.path.join(ROOT, "tests", "fuzz_oracle_lzma.py"))
fl = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fl)


def prepare(seed):
    d, variants = fl.make(seed)
    res = []
    for name, z in variants:
        cap = len(d) + 70000
        st, used, out = oracle.lzma_zip_decode(z, cap, -1)
        res.append((name, z, cap, st, used, len(out), zlib.crc32(out)))
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    with mp.Pool() as pool:                      # forks before the HIP context exists
        prep = pool.map(prepare, [seed * 1000003 + i for i in range(n)], chunksize=8)
    rows = [r for p in prep for r in p]
    import ctypes as C

    from tests import gpu_util
    from tests.test_gpu_lzma import run_lzma

    gpu_util.mz.require_gpu()
    L = gpu_util.mz.lib()
    L.mzhip_lzma_batch.restype = C.c_int32
    L.mzhip_lzma_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 5

    pays = [r[1] for r in rows]
    caps = [r[2] for r in rows]
    b, h_out, out_len, in_used, crc, status = run_lzma(gpu_util, pays, caps, [-1] * len(pays))
    bad = n_ok = 0
    for i, (name, z, cap, st, used, ol, k) in enumerate(rows):
        ok = (status[i] == 0) == (st == 0) and (st != 0 or (in_used[i], out_len[i], int(crc[i])) == (used, ol, k))
        n_ok += st == 0
        if not ok:
            bad += 1
            if bad < 10:
                print("MISMATCH", i, name, "gpu", status[i], in_used[i], out_len[i], hex(int(crc[i])), "oracle", st, used, ol, hex(k), z[:13].hex())
    print("gpu lzma fuzz: %d streams (%d decode, %d fail) -- %d mismatches" % (len(rows), n_ok, len(rows) - n_ok, bad))
    sys.exit(1 if bad else 0)

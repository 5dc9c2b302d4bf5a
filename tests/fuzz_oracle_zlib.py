#!/usr/bin/env python3
"""CPU-side probe (not a pytest): the oracle's restatement of inflate against the zlib of this interpreter on the streams of
tests/fuzz_gpu.py -- valid raw-DEFLATE streams of every level / strategy / window / memLevel with mid-stream flushes, and four
corruptions of each.  Compared: the error class (0 / Z_DATA_ERROR / Z_BUF_ERROR) of every stream, the bytes and the consumed
input of the ones that decode.  (Round 6: the device fuzz found two streams in 200 000 on which the restatement, not the device,
disagreed with zlib -- this run looks for more of them without a GPU.)
    python tests/fuzz_oracle_zlib.py [N=8000] [seed=1]"""
import importlib.util
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

_spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tests", "fuzz_gpu.py"))
fz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fz)


def check(args):
    z, cap = args
    st, used, out = oracle.inflate_raw(z, cap)
    d = zlib.decompressobj(-15)
    try:
        zo = d.decompress(z, cap + 1)
    except zlib.error as e:
        return st == -3, (st, used, len(out), "zlib: %s" % e)
    if not d.eof:
        if len(zo) > cap:  # more output than the capacity given to the oracle: its own verdict, not a stream property
            return st not in (0, -3, -5), (st, used, len(out), "zlib: output past the capacity")
        return st == -5, (st, used, len(out), "zlib: wants more input, %d bytes out" % len(zo))
    return st == 0 and out == zo and used == len(z) - len(d.unused_data), (st, used, len(out), "zlib: ok, %d bytes, %d unused" % (len(zo), len(d.unused_data)))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    c = synth.corpus()
    pays, caps = [], []
    for _ in range(n):
        d, z = fz.gen(rnd, c)
        pays.append(z + b"xyz")
        caps.append(len(d) + 16)
        for _ in range(4):
            pays.append(fz.corrupt(rnd, z))
            caps.append(len(d) + 70000)
    with mp.Pool() as pool:
        res = pool.map(check, list(zip(pays, caps)), chunksize=64)
    bad = 0
    for i, (ok, info) in enumerate(res):
        if not ok:
            bad += 1
            if bad < 10:
                print("MISMATCH", i, "oracle", info, pays[i][:48].hex(), len(pays[i]))
    print("oracle vs zlib %s: %d streams -- %d mismatches" % (zlib.ZLIB_RUNTIME_VERSION, len(pays), bad))
    sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""CPU-side probe (not a pytest; needs oracle/_ref): tests/test_oracle.py::test_xz_parity_with_reference at any size -- the
.xz restatement against the compiled reference (mz_stream_lzma READ, method 95) on corrupted / truncated variants of
synth.xz_cases(): the same accept / reject decision, on accept the same bytes and TOTAL_IN.
    python tests/fuzz_oracle_xz.py [N=20000] [seed=1]"""
import multiprocessing as mp
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

BASES = None


def init():
    global BASES
    BASES = [x for _, d, x in synth.xz_cases() if 0 < len(d) <= 100000]


def check(seed):
    rnd = random.Random(seed)
    ref = oracle.ref()
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
    st, used, out = oracle.xz_decode(x, 200000)
    if st == -109:
        return None  # filter chains outside the backend's scope
    r = ref.stream_decode(95, x, 200000)
    ok_ref = r["error"] in (0, 1) and r["rets"][-1] >= 0
    if (st == 0) != ok_ref:
        return (seed, k, "verdict", st, r["rets"][-2:], r["error"])
    if ok_ref and (out, used) != (r["out"], r["total_in"]):
        return (seed, k, "bytes / TOTAL_IN", used, r["total_in"], len(out), len(r["out"]))
    if not ok_ref and not (r["rets"][-1] == -3 and st in (-3, -5)):
        return (seed, k, "class", st, r["rets"][-2:])
    return None


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    with mp.Pool(initializer=init) as pool:
        res = pool.map(check, [seed * 1000003 + i for i in range(n)], chunksize=16)
    bad = [r for r in res if r]
    for b in bad[:10]:
        print("MISMATCH", b)
    print("oracle vs reference, method 95: %d cases -- %d mismatches" % (n, len(bad)))
    sys.exit(1 if bad else 0)
